#!/bin/bash
# the other BASELINE configurations on the final code (rank-47 scheme): config 5 on one GPU, its per-rank shape, the 8-rank line on virtual ranks
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05; mkdir -p $O
python bench.py --workload rect131072 --steps 20 --warmup 5 --no-cpu-baseline --no-api > $O/bench_rect131072_scheme47.json 2>/dev/null
python bench.py --workload leaf16384 --steps 500 --warmup 300 --no-cpu-baseline > $O/bench_leaf16384_scheme47.json 2>/dev/null
TAG=perrank python tools/time_product.py 16384 8192 131072 50 10 2>&1 | grep -v amdgpu.ids | tee $O/perrank_scheme47.log
( time python bench.py --gpus 8 --virtual-ranks --watchdog 200 > $O/bench_peer8_virtual_scheme47.json 2> $O/bench_peer8_virtual_scheme47.err ) 2>&1 | grep real
for f in bench_rect131072_scheme47 bench_leaf16384_scheme47 bench_peer8_virtual_scheme47; do python - <<PY
import json
d = json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
print("$f", round(d["ms_per_step"], 3), "%.4g" % d["value"], d.get("roofline", {}).get("lds", {}).get("frac"), d.get("verified", {}).get("matches_reference"), d.get("pipelined_ms_per_step"), d["config"].get("controller_wall_s"))
PY
done
