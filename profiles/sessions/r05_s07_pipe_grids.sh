#!/bin/bash
# block grids and block orders of the host pipeline at 65536^3, one box, two rounds
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05; mkdir -p $O
for round in 1 2; do
  for g in 2,2,2 4,2,2 4,2,1 2,4,2 4,4,2 8,2,2; do
    M4RI_AMD_PIPE_GRID=$g python tools/host_pipeline_grid_sweep.py 65536 2>&1 | grep -v amdgpu.ids
    M4RI_AMD_PIPE_ORDER=c M4RI_AMD_PIPE_GRID=$g python tools/host_pipeline_grid_sweep.py 65536 2>&1 | grep -v amdgpu.ids | sed 's/^grid/grid (columns of C outermost)/'
  done
done | tee $O/host_pipeline_grid_sweep_65536.log
M4RI_AMD_PIPE_ORDER=c M4RI_AMD_PIPE_GRID=4,2,2 python tools/host_pipeline_trace.py 65536 3 2>&1 | grep -v amdgpu.ids | tail -60 > $O/host_pipeline_timeline_65536_grid422_colmajor.log
