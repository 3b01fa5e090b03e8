#!/bin/bash
# round 3, closing run: the whole -m gpu suite, smoke, rocprofv3 passes over the default bench (tools/prof_bench.sh), the driver-style line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -q -m gpu > $O/final_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/final_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/final_smoke.log 2>&1
echo "smoke rc $?" >> $O/final_smoke.log
bash tools/prof_bench.sh r03_ > $O/final_prof_bench.log 2>&1
cp $R/gpurun_out/prof_bench/*.summary.txt $R/gpurun_out/prof_bench/bench_under_trace.json $R/gpurun_out/prof_bench/leaf_traffic.json $O/ 2>/dev/null
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/final_bench.json 2> $O/final_bench.err
echo "bench rc $?"
tail -4 $O/final_pytest_gpu.log; tail -2 $O/final_smoke.log; ls $O | grep -c summary; head -8 $O/trace.summary.txt; python - <<'PY'
import json,os
d=json.loads(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03/final_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d["roofline"]["lds"], d["verified"]["matches_reference"], d.get("api_ms"), d["api"])
PY
