#!/bin/bash
# round 4, GPU session 40: the whole -m gpu suite on the state with rows in blocks; default bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu > $O/s40_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/s40_pytest_gpu.log
tail -4 $O/s40_pytest_gpu.log
( time timeout 900 python bench.py ) > $O/s40_bench_default.json 2> $O/s40_bench_default.err
tail -4 $O/s40_bench_default.err; head -c 300 $O/s40_bench_default.json; echo
