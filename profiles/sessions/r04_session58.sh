#!/bin/bash
# round 4, GPU session 58: does the leaf launch's own split model pick well?  forced inner-dimension splits on single leaf launches
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
for shape in "16384 16384 16384" "8192 8192 8192" "4096 4096 4096" "12288 12288 12288" "20000 20000 20000" "32768 4096 32768" "4096 65536 4096" "8192 32768 8192" "16384 4096 16384" "24576 8192 8192"; do
  timeout 300 python tools/leaf_ksplit_sweep.py $shape >> $O/s58_ksplit.log 2>&1
done
grep "x" $O/s58_ksplit.log | grep ksplit
