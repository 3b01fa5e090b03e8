#!/bin/bash
# round 4, GPU session 4: the cost of a fresh result by strategy; distributed-matrix tests; bench peer on 8 virtual ranks; multi tests; threads
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
{ cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag; uname -r; nproc; free -g | head -2; } > $O/s4_host.txt 2>&1
for mode in populate huge lazy zero; do
  M4RI_AMD_STATS=1 M4RI_AMD_RESULT_CACHE=0 M4RI_AMD_FRESH=$mode timeout 300 python tools/fresh_result_timing.py >> $O/s4_fresh_result.log 2>&1
done
M4RI_AMD_STATS=1 timeout 300 python tools/fresh_result_timing.py >> $O/s4_fresh_result.log 2>&1
cat $O/s4_host.txt $O/s4_fresh_result.log
timeout 1500 python -m pytest tests/test_gpu_dmat.py -x -q -m gpu > $O/s4_pytest_dmat.log 2>&1
tail -15 $O/s4_pytest_dmat.log
timeout 900 python -m pytest tests/test_gpu_threads.py -x -q -m gpu > $O/s4_pytest_threads.log 2>&1
tail -5 $O/s4_pytest_threads.log
timeout 900 python bench.py --gpus 8 --transport peer --virtual-ranks --steps 3 --warmup 1 --no-cpu-baseline > $O/s4_bench_peer8_virtual.json 2> $O/s4_bench_peer8_virtual.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04/s4_bench_peer8_virtual.json"))
print(d["ms_per_step"], d["host_issue_ms_per_step"], d["config"]["schedule_stats"], d["config"]["timeline_ms_last_step"]["0"], d["config"]["timeline_ms_last_step"]["7"], d.get("verified"))
PY
timeout 2400 python -m pytest tests/test_gpu_multi.py -x -q -m gpu > $O/s4_pytest_multi.log 2>&1
tail -15 $O/s4_pytest_multi.log
