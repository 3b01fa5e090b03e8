#!/bin/bash
# round 4, GPU session 42: the multi-rank bench tests again after the bench.py change (row_blocks in the line)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_multi.py tests/test_gpu_dmat.py -x -q -m gpu > $O/s42_pytest_multi.log 2>&1
tail -4 $O/s42_pytest_multi.log
timeout 600 python bench.py --dims 65664,65664,65664 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-verify --no-api > $O/s42_bench_65664.json 2> $O/s42_bench_65664.err
head -c 1500 $O/s42_bench_65664.json; echo; tail -3 $O/s42_bench_65664.err
