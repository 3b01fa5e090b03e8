#!/bin/bash
# round 6, the final code: smoke, the driver-style bench lines, the rocprofv3 passes over the default bench, BASELINE configs 2 and 5 on one GPU,
# the complete N = 8 line on virtual ranks, the whole -m gpu suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=${OUT:-gpurun_out/r06}; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/bench65536_final.json 2> $O/bench65536_final.err; tail -c 300 $O/bench65536_final.err | grep -v amdgpu
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench65536_final_steps20.json 2>/dev/null
bash tools/prof_bench.sh r06 2>&1 | tail -2
python bench.py --workload rect131072 --steps 20 --warmup 5 --no-cpu-baseline --no-api > $O/bench_rect131072.json 2>/dev/null
python bench.py --workload leaf16384 --steps 500 --warmup 300 --no-cpu-baseline > $O/bench_leaf16384.json 2>/dev/null
( time python bench.py --gpus 8 --virtual-ranks --watchdog 200 > $O/bench_peer8_virtual_final.json 2> $O/bench_peer8_virtual_final.err ) 2>&1 | grep real
for f in bench65536_final bench65536_final_steps20 bench_rect131072 bench_leaf16384 bench_peer8_virtual_final; do python - <<PY
import json
d = json.loads([l for l in open("$O/$f.json").read().strip().splitlines() if l.startswith("{")][-1])
print("$f", round(d["ms_per_step"], 3), "%.4g" % d["value"], d.get("roofline", {}).get("lds", {}).get("frac"), d.get("verified", {}).get("matches_reference"), d.get("pipelined_ms_per_step"), d["config"].get("controller_wall_s"))
PY
done
( time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) 2>&1 | tee $O/pytest_gpu_full_final.log
