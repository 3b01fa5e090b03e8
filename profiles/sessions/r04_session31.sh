#!/bin/bash
# round 4, GPU session 31: full -m gpu suite with the depth model + RMW epilogue; bench lines for configs 2, 3, 5
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu > $O/s31_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/s31_pytest_gpu.log
tail -4 $O/s31_pytest_gpu.log
timeout 900 python bench.py --workload rect131072 --steps 10 --warmup 3 --no-cpu-baseline > $O/s31_bench_rect131072.json 2> $O/s31_bench_rect131072.err
head -c 400 $O/s31_bench_rect131072.json; echo
timeout 900 python bench.py --workload leaf16384 --steps 20 --warmup 5 --no-cpu-baseline > $O/s31_bench_leaf16384.json 2> $O/s31_bench_leaf16384.err
head -c 300 $O/s31_bench_leaf16384.json; echo
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/s31_bench65536.json 2> $O/s31_bench65536.err
head -c 300 $O/s31_bench65536.json; echo
