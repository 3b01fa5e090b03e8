#!/bin/bash
# round 4, GPU session 34: regression check of the callers one step up (PLE / TRSM at 65536) and of addmul after the epilogue change
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 900 python tools/l4_device_timing.py 32768 65536 > $O/s34_l4_device_timing.log 2>&1
cat $O/s34_l4_device_timing.log | tail -14
timeout 600 python tools/addmul_timing.py > $O/s34_addmul_timing.log 2>&1
cat $O/s34_addmul_timing.log
