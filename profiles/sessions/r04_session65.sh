#!/bin/bash
# round 4, GPU session 65: the default bench with config 1's adaptive stop rule in the cpu_baseline leg
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python bench.py ) > $O/s65_bench_default.json 2> $O/s65_bench_default.err
tail -4 $O/s65_bench_default.err; head -c 300 $O/s65_bench_default.json; echo
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04/s65_bench_default.json') if l.startswith('{"metric"')][-1])
print(json.dumps(d['cpu_baseline'].get('config1'))[:600])
PY
