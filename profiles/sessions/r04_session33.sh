#!/bin/bash
# round 4, GPU session 33: long soaks -- fused passes (random shapes / depths / strides), the fuzz test through the C ABI with other seeds, determinism
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
for seed in 21 22 23; do
  timeout 1500 python tools/fused_pass_soak.py 3000 $seed > $O/s33_fused_pass_soak_$seed.log 2>&1
  tail -1 $O/s33_fused_pass_soak_$seed.log
done
for seed in 31 32 33 34; do
  M4RI_AMD_FUZZ_SEED=$seed M4RI_AMD_FUZZ_CASES=4000 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k randomized > $O/s33_fuzz_$seed.log 2>&1
  tail -1 $O/s33_fuzz_$seed.log
done
M4RI_AMD_FUZZ_SEED=35 M4RI_AMD_FUZZ_CASES=600 M4RI_AMD_FUZZ_MAXDIM=6000 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k randomized > $O/s33_fuzz_big.log 2>&1
tail -1 $O/s33_fuzz_big.log
