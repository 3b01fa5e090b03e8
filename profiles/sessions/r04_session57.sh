#!/bin/bash
# round 4, GPU session 57: mzd_ple from host memory, cold against warm (the 250 ms of session 56 was a first call)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
ONLY=ple timeout 900 python tools/l4_device_timing.py 65536 65536 32768 65536 > $O/s57_l4_device_timing.log 2>&1
cat $O/s57_l4_device_timing.log | tail -10
