#!/bin/bash
# Strassen-Winograd at the top of the host block pipeline: tests, then the timelines of both schedules on one box, alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_gpu_host_pipeline.py tests/test_small_products.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r05/pytest_gpu_w7.log
for round in 1 2; do
  M4RI_AMD_PIPE_W7=0 python tools/host_pipeline_trace.py 65536 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/host_pipeline_timeline_65536_classical_$round.log
  M4RI_AMD_PIPE_W7=1 python tools/host_pipeline_trace.py 65536 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/host_pipeline_timeline_65536_w7_$round.log
  grep "^call" gpurun_out/r05/host_pipeline_timeline_65536_classical_$round.log | tr '\n' ' '; echo " <- classical"
  grep "^call" gpurun_out/r05/host_pipeline_timeline_65536_w7_$round.log | tr '\n' ' '; echo " <- w7"
done
tail -26 gpurun_out/r05/host_pipeline_timeline_65536_w7_2.log
