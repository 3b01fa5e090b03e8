#!/bin/bash
# round 4, GPU session 56: closing run -- the whole -m gpu suite with the thin-product test, the callers one step up, config 5 and config 2 lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu > $O/s56_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/s56_pytest_gpu.log
tail -4 $O/s56_pytest_gpu.log
timeout 900 python tools/l4_device_timing.py 65536 > $O/s56_l4_device_timing.log 2>&1
tail -7 $O/s56_l4_device_timing.log
timeout 900 python bench.py --workload rect131072 --steps 10 --warmup 3 --no-cpu-baseline > $O/s56_bench_rect131072.json 2> $O/s56_bench_rect131072.err
head -c 300 $O/s56_bench_rect131072.json; echo
python -c "import __graft_entry__ as g; g.smoke()" > $O/s56_smoke.log 2>&1; tail -2 $O/s56_smoke.log
