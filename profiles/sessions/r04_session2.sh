#!/bin/bash
# round 4, GPU session 2: power traces on the RIGHT card (HIP device 0 by PCI address), the fresh-result path (mzd_mul(NULL, ...))
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
for v in base nobarrier nobuild nobuild_nobarrier; do
  timeout 300 python tools/power_trace.py --smi --hz 100 --out $O/power --tag leaf_$v -- build/leaf_check_$v --one 32 1 11 343 150 > $O/s2_power_leaf_$v.log 2>&1
done
timeout 300 python tools/power_trace.py --smi --hz 100 --out $O/power --tag passes -- python tools/passes_only.py 65536 2500 > $O/s2_power_passes.log 2>&1
timeout 300 python tools/power_trace.py --smi --hz 100 --out $O/power --tag product65536 -- python tools/prof_product.py 65536 65536 65536 100 > $O/s2_power_product.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_pipeline.py tests/test_dropin_preload.py tests/test_cabi.py -x -q -m gpu > $O/s2_pytest_subset.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-traffic > $O/s2_bench65536_api.json 2> $O/s2_bench65536_api.err
for f in $O/s2_power_*.log; do head -c 900 $f; echo; done; tail -3 $O/s2_pytest_subset.log; python - <<'PY'
import json
d=json.load(open("gpurun_out/r04/s2_bench65536_api.json"))
print(d["ms_per_step"], d.get("api"))
PY
