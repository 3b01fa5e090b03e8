#!/bin/bash
# round 4, GPU session 16: which depth should the engine pick by itself, over a range of shapes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/depth_rule_sweep.py > $O/s16_depth_rule_sweep.log 2>&1
grep -v amdgpu.ids $O/s16_depth_rule_sweep.log
