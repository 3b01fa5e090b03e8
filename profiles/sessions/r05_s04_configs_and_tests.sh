#!/bin/bash
# new tests of the round (small-product routing at the default threshold, poisoned result blocks, pins, threads), BASELINE configs 2 and 5 on
# one box, the per-rank shape of config 5's 8-GPU row-slab schedule, the depth sweep of the five BASELINE-derived shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_small_products.py tests/test_gpu_host_pipeline.py tests/test_gpu_residency.py tests/test_gpu_threads.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r05/pytest_gpu_advice.log
python bench.py --workload leaf16384 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r05/bench_leaf16384_w10.json 2>/dev/null
python bench.py --workload leaf16384 --steps 500 --warmup 300 --no-cpu-baseline > gpurun_out/r05/bench_leaf16384.json 2>/dev/null
python tools/leaf_ksplit_sweep.py 16384 16384 16384 50 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/leaf_ksplit_sweep_16384.log
python bench.py --workload rect131072 --steps 20 --warmup 5 --no-cpu-baseline --no-api > gpurun_out/r05/bench_rect131072.json 2>/dev/null
python tools/depth_model_sweep.py 65536,65536,65536 32768,32768,32768 16384,16384,16384 131072,8192,131072 16384,8192,131072 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/depth_model_box.log
rocm-smi --showproductname --showserial 2>/dev/null | head -12 >> gpurun_out/r05/depth_model_box.log
for f in leaf16384_w10 leaf16384 rect131072; do python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r05/bench_$f.json") if l.startswith("{")][-1])
print("$f", "ms_per_step", round(d["ms_per_step"],4), "launch_ms", round(d["roofline"]["launch_ms"],4), "lds.frac", d["roofline"]["lds"] and round(d["roofline"]["lds"]["frac"],4), "levels", d["config"]["strassen_levels"])
PY
done
