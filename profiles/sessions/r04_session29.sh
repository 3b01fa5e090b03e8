#!/bin/bash
# round 4, GPU session 29: by-kernel split of a ragged shape at depth 0 / 1 / 2 (where do the peel strips cost?)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in 0 1 2; do
  ( cd /tmp; M4RI_AMD_LEVELS=$L rocprofv3 --kernel-trace --stats -d $R/$O/tr29 -o t -- python $R/tools/prof_product.py 50000 12000 90000 5 > $R/$O/s29_trace_L$L.log 2>&1
    python $R/tools/rocpd_summary.py $(find $R/$O/tr29 -name "*results.db" | head -1) > $R/$O/s29_trace_L$L.summary.txt 2>&1; rm -rf $R/$O/tr29 )
  grep shape $O/s29_trace_L$L.log
  head -16 $O/s29_trace_L$L.summary.txt
done
