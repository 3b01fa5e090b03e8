#!/bin/bash
# round 4, GPU session 51: the per-rank products of the 2 / 4 / 8-GPU schedules on the final code (tools/rank_shapes_timing.py)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 900 python tools/rank_shapes_timing.py > $O/s51_rank_shapes_timing.log 2>&1
cat $O/s51_rank_shapes_timing.log | cut -c1-200
