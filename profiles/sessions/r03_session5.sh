#!/bin/bash
# round 3, GPU session 5: the whole -m gpu suite again (after the bench.py fix), rank shapes incl. the 2x2 units
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r03/s5_pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03/s5_pytest_gpu.log
timeout 600 python tools/rank_shapes_timing.py > gpurun_out/r03/s5_rank_shapes.log 2>&1
tail -5 gpurun_out/r03/s5_pytest_gpu.log; tail -16 gpurun_out/r03/s5_rank_shapes.log
