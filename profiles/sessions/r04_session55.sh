#!/bin/bash
# round 4, GPU session 55: final state -- the whole -m gpu suite, the plans whose model changed, default bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu > $O/s55_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/s55_pytest_gpu.log
tail -4 $O/s55_pytest_gpu.log
timeout 1500 python tools/depth_model_sweep.py 50000,12000,90000 40977,16384,16384 69632,8192,131072 20480,20480,20480 100003,50021,70017 70000,70000,70000 66048,65536,65536 33768,32768,32768 > $O/s55_sweep.log 2>&1
cut -c1-330 $O/s55_sweep.log
( time timeout 900 python bench.py ) > $O/s55_bench_default.json 2> $O/s55_bench_default.err
tail -4 $O/s55_bench_default.err; head -c 300 $O/s55_bench_default.json; echo
timeout 600 python tools/row_blocks_soak.py 80 9 > $O/s55_row_blocks_soak.log 2>&1; tail -1 $O/s55_row_blocks_soak.log
