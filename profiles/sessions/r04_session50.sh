#!/bin/bash
# round 4, GPU session 50: soaks on the shipped state -- plans on big ragged shapes against one product at another depth; the fuzz test with large dimensions; fused passes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 1500 python tools/row_blocks_soak.py 150 3 > $O/s50_row_blocks_soak.log 2>&1
tail -2 $O/s50_row_blocks_soak.log
M4RI_AMD_FUZZ_SEED=41 M4RI_AMD_FUZZ_CASES=150 M4RI_AMD_FUZZ_MAXDIM=9000 timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k randomized > $O/s50_fuzz_big.log 2>&1
tail -1 $O/s50_fuzz_big.log
timeout 900 python tools/fused_pass_soak.py 2000 51 > $O/s50_fused_pass_soak.log 2>&1
tail -1 $O/s50_fused_pass_soak.log
