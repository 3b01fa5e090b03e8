#!/bin/bash
# round 4, GPU session 38: by-kernel split of 34000 x 20000 x 20000 as two row blocks (where the model is 13 % optimistic) and as one product
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rb in 1 0; do
( cd /tmp; M4RI_AMD_ROW_BLOCKS=$rb rocprofv3 --kernel-trace --stats -d $R/$O/tr38 -o t -- python $R/tools/prof_product.py 34000 20000 20000 10 > $R/$O/s38_trace_rb$rb.log 2>&1
  python $R/tools/rocpd_summary.py $(find $R/$O/tr38 -name "*results.db" | head -1) > $R/$O/s38_trace_rb$rb.summary.txt 2>&1; rm -rf $R/$O/tr38 )
grep shape $O/s38_trace_rb$rb.log
head -16 $O/s38_trace_rb$rb.summary.txt
done
