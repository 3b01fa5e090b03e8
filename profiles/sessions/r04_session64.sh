#!/bin/bash
# round 4, GPU session 64: the closing profile set from one box on the final binary -- default bench, --steps 20, the rocprofv3 passes, configs 2 and 5
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python bench.py ) > $O/s64_bench_default.json 2> $O/s64_bench_default.err
tail -4 $O/s64_bench_default.err; head -c 300 $O/s64_bench_default.json; echo
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/s64_bench_steps20.json 2> $O/s64_bench_steps20.err
head -c 300 $O/s64_bench_steps20.json; echo
bash tools/prof_bench.sh r04 > $O/s64_prof_bench.log 2>&1
head -8 gpurun_out/prof_bench/trace.summary.txt
timeout 900 python bench.py --workload rect131072 --steps 10 --warmup 3 --no-cpu-baseline > $O/s64_bench_rect131072.json 2> $O/s64_bench_rect131072.err
head -c 300 $O/s64_bench_rect131072.json; echo
timeout 900 python bench.py --workload leaf16384 --steps 50 --warmup 10 --no-cpu-baseline > $O/s64_bench_leaf16384.json 2> $O/s64_bench_leaf16384.err
head -c 300 $O/s64_bench_leaf16384.json; echo
