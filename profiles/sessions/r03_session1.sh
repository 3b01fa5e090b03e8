#!/bin/bash
# round 3, GPU session 1: the new multi-rank tests, the bench line, rank shapes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_multi.py -x -q -m gpu -k "bench or rccl" > gpurun_out/r03/s1_pytest_multi.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03/s1_pytest_multi.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/s1_bench.json 2> gpurun_out/r03/s1_bench.err
timeout 600 python tools/rank_shapes_timing.py > gpurun_out/r03/s1_rank_shapes.log 2>&1
tail -5 gpurun_out/r03/s1_pytest_multi.log; tail -c 1500 gpurun_out/r03/s1_bench.json; tail -8 gpurun_out/r03/s1_rank_shapes.log
