#!/bin/bash
# round 4, GPU session 45: partly filled tiles on the gather-only waves with three rows in flight -- parity, full-size timing, thin and half-tile shapes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/s45_pytest.log 2>&1
tail -3 $O/s45_pytest.log
for rep in 1 2 3; do timeout 300 python tools/prof_product.py 65536 65536 65536 8 >> $O/s45_timing.log 2>&1; done
timeout 300 python tools/prof_product.py 128 65664 65664 10 >> $O/s45_timing.log 2>&1
timeout 300 python tools/prof_product.py 464 66000 66000 10 >> $O/s45_timing.log 2>&1
timeout 300 python tools/prof_product.py 1699 50021 70017 10 >> $O/s45_timing.log 2>&1
timeout 300 python tools/prof_product.py 4464 70000 70000 10 >> $O/s45_timing.log 2>&1
timeout 300 python tools/prof_product.py 2048 65536 65536 10 >> $O/s45_timing.log 2>&1
grep shape $O/s45_timing.log
timeout 900 python tools/depth_model_sweep.py 24576,24576,24576 12288,12288,12288 36864,36864,36864 16384,65536,65536 20480,20480,20480 40960,40960,40960 49152,49152,49152 65664,65664,65664 70000,70000,70000 > $O/s45_sweep.log 2>&1
cut -c1-420 $O/s45_sweep.log
