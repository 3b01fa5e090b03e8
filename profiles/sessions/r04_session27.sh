#!/bin/bash
# round 4, GPU session 27: BASELINE config 5's shape (131072 x 8192 x 131072) one level deeper than the rule (leaves of 2048 inner bits)?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
for rep in 1 2; do
  timeout 300 python tools/prof_product.py 131072 8192 131072 5 >> $O/s27_timing.log 2>&1
  timeout 300 python tools/prof_product.py 131072 8192 131072 5 2048 >> $O/s27_timing.log 2>&1
done
timeout 300 python tools/prof_product.py 131072 8192 131072 5 1024 >> $O/s27_timing.log 2>&1
timeout 300 python tools/prof_product.py 65536 8192 65536 5 >> $O/s27_timing.log 2>&1
timeout 300 python tools/prof_product.py 65536 8192 65536 5 2048 >> $O/s27_timing.log 2>&1
timeout 300 python tools/prof_product.py 32768 4096 32768 10 >> $O/s27_timing.log 2>&1
timeout 300 python tools/prof_product.py 32768 4096 32768 10 2048 >> $O/s27_timing.log 2>&1
grep "shape\|Error\|error" $O/s27_timing.log
