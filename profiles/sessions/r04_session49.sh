#!/bin/bash
# round 4, GPU session 49: the whole -m gpu suite on the shipped leaf (rows of partly filled tiles on the gather-only waves), default bench, leaf-only bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu > $O/s49_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/s49_pytest_gpu.log
tail -4 $O/s49_pytest_gpu.log
( time timeout 900 python bench.py ) > $O/s49_bench_default.json 2> $O/s49_bench_default.err
tail -4 $O/s49_bench_default.err; head -c 300 $O/s49_bench_default.json; echo
timeout 900 python bench.py --workload leaf16384 --steps 50 --warmup 5 --no-cpu-baseline > $O/s49_bench_leaf16384.json 2> $O/s49_bench_leaf16384.err
head -c 300 $O/s49_bench_leaf16384.json; echo
timeout 300 python tools/stress_determinism.py > $O/s49_stress_determinism.log 2>&1; tail -3 $O/s49_stress_determinism.log
