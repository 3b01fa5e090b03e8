#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03
mkdir -p $O; cd $R; export TMPDIR=/tmp
bash tools/prof_bench.sh r03_ > $O/final_prof_bench.log 2>&1
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/final_bench.json 2> $O/final_bench.err
head -9 $R/gpurun_out/prof_bench/trace.summary.txt; cat $R/gpurun_out/prof_bench/leaf_traffic.json | head -5
python - <<'PY'
import json,os
d=json.loads(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03/final_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d["roofline"]["lds"]["frac"], d["roofline"]["lds"].get("frac_in_cycles"), d["api_ms"], d["api"]["c_null_ms_min"])
PY
