#!/bin/bash
# round 4, GPU session 39: the refined model's plans on shapes whose plan changed, and on new ones
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 1500 python tools/depth_model_sweep.py 36864,36864,36864 73728,16384,65536 28672,28672,28672 45056,45056,45056 61440,61440,61440 66000,66000,66000 34000,20000,20000 \
    36900,20000,20000 40977,16384,16384 98304,32768,32768 20480,65536,65536 12288,65536,65536 69632,8192,131072 > $O/s39_row_blocks_sweep2.log 2>&1
cut -c1-330 $O/s39_row_blocks_sweep2.log
