#!/bin/bash
# round 6, session 3: the measurements behind the second host routine (small_host.cpp) and the small leaf (m4rm_small.hip)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r06s3; mkdir -p $O
# the host routine against the reference, from C (no GPU needed), and against the GPU path through mzd_mul
cc -O2 -o /tmp/small_host_timing tests/small_host_timing.c -ldl && /tmp/small_host_timing . > $O/small_host_timing.log 2>&1
python tools/small_crossover.py 2>&1 | grep -v amdgpu.ids > $O/small_crossover.log
python tests/crossover_cpu_gpu.py 2>&1 | grep -v amdgpu.ids > $O/crossover_cpu_gpu.log
# device-resident small products: the small leaf never / by the engine's rule (same box, same binary), extreme shapes, forced inner splits
M4RI_AMD_SMALL_LEAF=0 python tools/small_leaf_timing.py never 2>&1 | grep -v amdgpu.ids > $O/small_leaf_never.log
python tools/small_leaf_timing.py rule 2>&1 | grep -v amdgpu.ids > $O/small_leaf_rule.log
M4RI_AMD_SMALL_LEAF=0 python tools/extreme_shapes_small_leaf.py 2>&1 | grep -v amdgpu.ids > $O/extreme_never.log
python tools/extreme_shapes_small_leaf.py 2>&1 | grep -v amdgpu.ids > $O/extreme_rule.log
for k in 1 2 4 8; do M4RI_AMD_SMALL_KS=$k python tools/small_leaf_timing.py ks$k 2>&1 | grep -v amdgpu.ids | head -12; done > $O/small_leaf_forced_splits.log
# the callers one step up, and why the first block products of a host call are slow
for s in 0 auto; do echo "== M4RI_AMD_SMALL_LEAF=$s"; if [ $s = auto ]; then python tools/l4_device_timing.py 1024 2048 4096 8192 16384 32768 65536 2>&1 | grep resident; else M4RI_AMD_SMALL_LEAF=$s python tools/l4_device_timing.py 1024 2048 4096 8192 16384 32768 65536 2>&1 | grep resident; fi; done > $O/l4_small_leaf.log
python tools/upload_interference_probe.py 2>&1 | grep -v amdgpu.ids > $O/upload_interference.log
# parity: the new tests, then the whole suite and the soaks (sessions/r06_s11_soak_small.sh, r06_s05_final.sh)
python -m pytest tests/test_gpu_small_leaf.py tests/test_small_products.py tests/test_gpu_batch.py -q -m gpu 2>&1 | tail -2
tail -3 $O/small_host_timing.log; tail -3 $O/small_leaf_rule.log | cut -c1-160
