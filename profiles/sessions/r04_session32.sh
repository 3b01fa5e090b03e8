#!/bin/bash
# round 4, GPU session 32: soak of the fused passes (random shapes, depths, strides, accumulate) + the fuzz test with more cases and other seeds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 1200 python tools/fused_pass_soak.py 300 7 > $O/s32_fused_pass_soak.log 2>&1
tail -3 $O/s32_fused_pass_soak.log; grep -c MISMATCH $O/s32_fused_pass_soak.log
for seed in 11 12; do
  M4RI_AMD_FUZZ_SEED=$seed M4RI_AMD_FUZZ_CASES=400 timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k randomized > $O/s32_fuzz_$seed.log 2>&1
  tail -2 $O/s32_fuzz_$seed.log
done
