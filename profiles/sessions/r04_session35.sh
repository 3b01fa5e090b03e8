#!/bin/bash
# round 4, GPU session 35: rows in blocks -- the engine's plan against every single-product depth on shapes whose rows do not tile; parity of ragged shapes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mul or fuzz or randomized or ragged or window" > $O/s35_pytest.log 2>&1
tail -3 $O/s35_pytest.log
timeout 1500 python tools/depth_model_sweep.py 65664,65664,65664 36864,36864,36864 40960,40960,40960 100003,50021,70017 50000,12000,90000 70000,70000,70000 45000,45000,45000 \
   20480,20480,20480 49152,49152,49152 30000,30000,30000 16421,16453,16523 57344,57344,57344 24576,24576,24576 60000,60000,60000 50000,50000,50000 20000,20000,20000 \
   69632,65536,65536 73728,16384,65536 33000,33000,33000 9000,9000,9000 > $O/s35_row_blocks_sweep.log 2>&1
cat $O/s35_row_blocks_sweep.log
