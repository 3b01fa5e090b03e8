#!/bin/bash
# round 4, GPU session 25: where the host pipeline's time goes (timeline of mzd_mul on host matrices at 65536^3 and 32768^3)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 600 python tools/host_pipeline_trace.py 65536 3 > $O/s25_pipe65536.log 2>&1
timeout 600 python tools/host_pipeline_trace.py 32768 3 > $O/s25_pipe32768.log 2>&1
tail -40 $O/s25_pipe65536.log
