#!/bin/bash
# round 4, GPU session 1: what power / clock / temperature telemetry the box offers; power traces of the leaf (as shipped, no
# barrier, no table building) and of the passes; configs 2 and 5 back on the record; the headline on this box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
{ ls /sys/class/drm/; ls /sys/class/drm/card*/device/hwmon/hwmon*/; cat /sys/class/drm/card*/device/hwmon/hwmon*/*_label 2>/dev/null; which amd-smi rocm-smi;
  timeout 60 amd-smi static --limit 2>&1 | head -40; timeout 60 amd-smi metric --throttle 2>&1 | head -60; timeout 60 amd-smi metric --power --clock --temperature 2>&1 | head -80; } > $O/s1_telemetry.txt 2>&1
for v in base nobarrier nobuild; do
  timeout 300 python tools/power_trace.py --smi --hz 100 --out $O/power --tag leaf_$v -- build/leaf_check_$v --one 32 1 11 343 120 > $O/s1_power_leaf_$v.log 2>&1
done
timeout 300 python tools/power_trace.py --smi --hz 100 --out $O/power --tag passes -- python tools/passes_only.py 65536 1500 > $O/s1_power_passes.log 2>&1
timeout 300 python bench.py --workload leaf16384 --steps 50 --warmup 5 --no-cpu-baseline > $O/s1_bench_leaf16384.json 2> $O/s1_bench_leaf16384.err
timeout 600 python bench.py --workload rect131072 --steps 10 --warmup 2 --no-cpu-baseline > $O/s1_bench_rect131072.json 2> $O/s1_bench_rect131072.err
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-api > $O/s1_bench65536.json 2> $O/s1_bench65536.err
tail -3 $O/s1_power_leaf_*.log $O/s1_power_passes.log; head -c 600 $O/s1_bench_leaf16384.json; echo; head -c 400 $O/s1_bench65536.json; tail -30 $O/s1_telemetry.txt
