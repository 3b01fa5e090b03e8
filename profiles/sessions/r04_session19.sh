#!/bin/bash
# round 4, GPU session 19: the final state (default depth 4) -- the whole -m gpu suite, the bench profile set, bench lines of all three workloads
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu > $O/s19_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/s19_pytest_gpu.log
tail -5 $O/s19_pytest_gpu.log
( time timeout 900 python bench.py ) > $O/s19_bench_default.json 2> $O/s19_bench_default.err
tail -4 $O/s19_bench_default.err; head -c 300 $O/s19_bench_default.json; echo
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/s19_bench_steps20.json 2> $O/s19_bench_steps20.err
head -c 300 $O/s19_bench_steps20.json; echo
timeout 300 python bench.py --workload leaf16384 --steps 50 --warmup 5 --no-cpu-baseline > $O/s19_bench_leaf16384.json 2> $O/s19_bench_leaf16384.err
timeout 600 python bench.py --workload rect131072 --steps 10 --warmup 2 --no-cpu-baseline > $O/s19_bench_rect131072.json 2> $O/s19_bench_rect131072.err
head -c 200 $O/s19_bench_leaf16384.json; echo; head -c 200 $O/s19_bench_rect131072.json; echo
bash tools/prof_bench.sh r04 > $O/s19_prof_bench.log 2>&1
tail -3 $O/s19_prof_bench.log
timeout 900 python bench.py --gpus 8 --transport peer --virtual-ranks --steps 4 --warmup 2 --no-cpu-baseline > $O/s19_bench_peer8_virtual.json 2> $O/s19_bench_peer8_virtual.err
head -c 250 $O/s19_bench_peer8_virtual.json; echo
