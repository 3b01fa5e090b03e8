#!/bin/bash
# round 4, GPU session 26: by-kernel split of the smaller cubes (16384^3 = BASELINE config 2's shape, 32768^3 = the sharded sub-product)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 16384 32768 8192; do
  timeout 300 python tools/prof_product.py $n $n $n 20 >> $O/s26_timing.log 2>&1
  ( cd /tmp; rocprofv3 --kernel-trace --stats -d $R/$O/tr26 -o t -- python $R/tools/prof_product.py $n $n $n 20 > $R/$O/s26_trace_$n.log 2>&1
    python $R/tools/rocpd_summary.py $(find $R/$O/tr26 -name "*results.db" | head -1) > $R/$O/s26_trace_$n.summary.txt 2>&1; rm -rf $R/$O/tr26 )
  head -12 $O/s26_trace_$n.summary.txt
done
grep shape $O/s26_timing.log
