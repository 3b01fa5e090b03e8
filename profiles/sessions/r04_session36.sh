#!/bin/bash
# round 4, GPU session 36: by-kernel split of the row-block plan of 50000 x 12000 x 90000 (where the model is 1 ms optimistic)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp; rocprofv3 --kernel-trace --stats -d $R/$O/tr36 -o t -- python $R/tools/prof_product.py 50000 12000 90000 5 > $R/$O/s36_trace.log 2>&1
  python $R/tools/rocpd_summary.py $(find $R/$O/tr36 -name "*results.db" | head -1) > $R/$O/s36_trace.summary.txt 2>&1; rm -rf $R/$O/tr36 )
grep shape $O/s36_trace.log
head -20 $O/s36_trace.summary.txt
( cd /tmp; M4RI_AMD_ROW_BLOCKS=0 rocprofv3 --kernel-trace --stats -d $R/$O/tr36 -o t -- python $R/tools/prof_product.py 50000 12000 90000 5 > $R/$O/s36_trace_single.log 2>&1
  python $R/tools/rocpd_summary.py $(find $R/$O/tr36 -name "*results.db" | head -1) > $R/$O/s36_trace_single.summary.txt 2>&1; rm -rf $R/$O/tr36 )
grep shape $O/s36_trace_single.log
head -8 $O/s36_trace_single.summary.txt
