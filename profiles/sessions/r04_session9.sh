#!/bin/bash
# round 4, GPU session 9: small-product crossover (reference vs GPU path vs host routine); the whole -m gpu suite; the driver's bench command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tests/crossover_cpu_gpu.py > $O/s9_crossover_cpu_gpu.log 2>&1
head -24 $O/s9_crossover_cpu_gpu.log
timeout 3000 python -m pytest tests -x -q -m gpu > $O/s9_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/s9_pytest_gpu.log
tail -6 $O/s9_pytest_gpu.log
( time timeout 900 python bench.py ) > $O/s9_bench_default.json 2> $O/s9_bench_default.err
tail -4 $O/s9_bench_default.err; head -c 600 $O/s9_bench_default.json
