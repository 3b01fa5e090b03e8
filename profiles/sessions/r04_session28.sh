#!/bin/bash
# round 4, GPU session 28: every shape at every Strassen depth (the data for the depth rule)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 1500 python tools/depth_model_sweep.py > $O/s28_depth_model_sweep.log 2>&1
cat $O/s28_depth_model_sweep.log
