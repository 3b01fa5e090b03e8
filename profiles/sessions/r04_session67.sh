#!/bin/bash
# round 4, GPU session 67: the opt-in 262144^3 product (8 GiB per matrix, six levels, the top ones depth-first) against the reference's fingerprint, on the final code
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
M4RI_AMD_HUGE=1 timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "262144 or 131072_cubed" --durations=5 > $O/s67_pytest_huge.log 2>&1
tail -12 $O/s67_pytest_huge.log
timeout 600 python tools/prof_product.py 262144 262144 262144 2 > $O/s67_timing.log 2>&1; grep shape $O/s67_timing.log
