#!/bin/bash
# round 4, GPU session 20: fused-pass parity incl. the LDS up pass on small shapes; a longer fuzz of the entry points through the GPU path
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "four_level or three_level" > $O/s20_pytest_fused.log 2>&1
tail -3 $O/s20_pytest_fused.log
M4RI_AMD_FUZZ_CASES=600 M4RI_AMD_FUZZ_SEED=4242 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "randomized" > $O/s20_pytest_fuzz.log 2>&1
tail -3 $O/s20_pytest_fuzz.log
M4RI_AMD_FUZZ_CASES=300 M4RI_AMD_FUZZ_SEED=99 M4RI_AMD_FUZZ_MAXDIM=3000 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "randomized" > $O/s20_pytest_fuzz2.log 2>&1
tail -3 $O/s20_pytest_fuzz2.log
