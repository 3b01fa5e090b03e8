#!/bin/bash
# round 4, GPU session 63: closing run after the split-model change -- the whole -m gpu suite, default bench, --steps 20, configs 2 and 5, 16384^3 and 32768^3 products
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu > $O/s63_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/s63_pytest_gpu.log
tail -4 $O/s63_pytest_gpu.log
( time timeout 900 python bench.py ) > $O/s63_bench_default.json 2> $O/s63_bench_default.err
tail -4 $O/s63_bench_default.err; head -c 300 $O/s63_bench_default.json; echo
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/s63_bench_steps20.json 2> $O/s63_bench_steps20.err
head -c 300 $O/s63_bench_steps20.json; echo
timeout 900 python bench.py --workload rect131072 --steps 10 --warmup 3 --no-cpu-baseline > $O/s63_bench_rect131072.json 2> $O/s63_bench_rect131072.err
head -c 300 $O/s63_bench_rect131072.json; echo
timeout 900 python bench.py --workload leaf16384 --steps 50 --warmup 10 --no-cpu-baseline > $O/s63_bench_leaf16384.json 2> $O/s63_bench_leaf16384.err
head -c 300 $O/s63_bench_leaf16384.json; echo
timeout 300 python tools/prof_product.py 16384 16384 16384 50 >> $O/s63_timing.log 2>&1
timeout 300 python tools/prof_product.py 32768 32768 32768 20 >> $O/s63_timing.log 2>&1
grep shape $O/s63_timing.log
