#!/bin/bash
# round 4, GPU session 14: closing run -- the whole -m gpu suite, smoke(), the driver's bench command, race hunts
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu > $O/s14_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/s14_pytest_gpu.log
tail -6 $O/s14_pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/s14_smoke.log 2>&1; tail -2 $O/s14_smoke.log
( time timeout 900 python bench.py ) > $O/s14_bench_default.json 2> $O/s14_bench_default.err
tail -4 $O/s14_bench_default.err; head -c 400 $O/s14_bench_default.json; echo
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/s14_bench_steps20.json 2> $O/s14_bench_steps20.err
head -c 300 $O/s14_bench_steps20.json; echo
timeout 900 python tools/stress_determinism.py > $O/s14_stress_determinism.log 2>&1; tail -3 $O/s14_stress_determinism.log
for k in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_dmat.py tests/test_gpu_threads.py -x -q -m gpu 2>&1 | tail -1; done > $O/s14_dmat_threads_x3.log 2>&1; cat $O/s14_dmat_threads_x3.log
