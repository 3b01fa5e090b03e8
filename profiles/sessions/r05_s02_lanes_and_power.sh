#!/bin/bash
# the lanes / staged-pairs / link-probe tests of the multi-device path, then the 100 Hz power traces of the serial schedule and of
# "the three passes once more under the leaf" (same box, alternating)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r05/overlap_power
timeout 1500 python -m pytest tests/test_gpu_dmat.py -x -q 2>&1 | tail -15 | tee gpurun_out/r05/pytest_gpu_dmat.log
for round in 1 2; do
  TAG="base" python tools/power_trace.py --hz 100 --out gpurun_out/r05/overlap_power --tag serial_$round -- python tools/time_product.py 65536 65536 65536 150 20 2>&1 | grep -v amdgpu.ids
  TAG="passes-under-leaf (all 3)" M4RI_AMD_OVERLAP_EXP=7 python tools/power_trace.py --hz 100 --out gpurun_out/r05/overlap_power --tag passes_under_leaf_$round -- python tools/time_product.py 65536 65536 65536 150 20 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee gpurun_out/r05/overlap_power/runs.log
