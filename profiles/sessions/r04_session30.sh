#!/bin/bash
# round 4, GPU session 30: the depth model on shapes it was not fitted to; the read-modify-write epilogue of unsplit addmul leaves
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "addmul or window or fuzz or remainder or ragged" > $O/s30_pytest.log 2>&1
tail -3 $O/s30_pytest.log
timeout 1500 python tools/depth_model_sweep.py 16421,16453,16523 100003,50021,70017 50000,12000,90000 16384,8192,131072 16384,65536,32768 32768,32768,16384 65536,65536,1024 \
   30000,30000,30000 45000,45000,45000 36864,36864,36864 57344,57344,57344 65664,65664,65664 65536,32768,65536 131072,32768,32768 24576,8192,49152 70000,524288,512 \
   8192,8192,8192 12288,12288,12288 20480,20480,20480 40960,40960,40960 131072,8192,131072 131072,16384,131072 > $O/s30_depth_model_validation.log 2>&1
cat $O/s30_depth_model_validation.log
