#!/bin/bash
# round 4, GPU session 43: the host pipeline's row cuts on the coarse grid -- parity, and the timeline of a 65664^3 product from host memory
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_host_pipeline.py -x -q -m gpu > $O/s43_pytest.log 2>&1
tail -3 $O/s43_pytest.log
timeout 600 python tools/host_pipeline_trace.py 65664 3 > $O/s43_pipe65664.log 2>&1
tail -24 $O/s43_pipe65664.log
timeout 600 python tools/host_pipeline_trace.py 65536 3 > $O/s43_pipe65536.log 2>&1
grep "call" $O/s43_pipe65536.log
