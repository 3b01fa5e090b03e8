#!/bin/bash
# round 4, GPU session 22: final state again (B-side LDS down pass the default) -- the whole -m gpu suite, bench lines, the profile set, the depth sweep
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu > $O/s22_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/s22_pytest_gpu.log
tail -4 $O/s22_pytest_gpu.log
( time timeout 900 python bench.py ) > $O/s22_bench_default.json 2> $O/s22_bench_default.err
tail -4 $O/s22_bench_default.err; head -c 300 $O/s22_bench_default.json; echo
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/s22_bench_steps20.json 2> $O/s22_bench_steps20.err
head -c 300 $O/s22_bench_steps20.json; echo
bash tools/prof_bench.sh r04 > $O/s22_prof_bench.log 2>&1
head -8 gpurun_out/prof_bench/trace.summary.txt
for rep in 1 2; do
  timeout 300 python tools/prof_product.py 65536 65536 65536 8 >> $O/s22_timing.log 2>&1
  timeout 300 python tools/prof_product.py 65536 65536 65536 8 8192 3 >> $O/s22_timing.log 2>&1
done
grep shape $O/s22_timing.log
