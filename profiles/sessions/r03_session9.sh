#!/bin/bash
# round 3, GPU session 9: the multi-rank bench paths incl. config 4 at full size over 8 ranks
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 3000 python -m pytest tests/test_gpu_multi.py -x -q -m gpu --durations=8 > gpurun_out/r03/s9_pytest_multi.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03/s9_pytest_multi.log
tail -16 gpurun_out/r03/s9_pytest_multi.log
