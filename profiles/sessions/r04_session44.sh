#!/bin/bash
# round 4, GPU session 44: by-kernel split of 65664^3 (65536 rows at four levels + 128 rows + strips): what do the thin pieces cost?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp; rocprofv3 --kernel-trace --stats -d $R/$O/tr44 -o t -- python $R/tools/prof_product.py 65664 65664 65664 5 > $R/$O/s44_trace.log 2>&1
  python $R/tools/rocpd_summary.py $(find $R/$O/tr44 -name "*results.db" | head -1) > $R/$O/s44_trace.summary.txt 2>&1; rm -rf $R/$O/tr44 )
grep shape $O/s44_trace.log
head -20 $O/s44_trace.summary.txt
timeout 300 python tools/prof_product.py 128 65664 65664 10 >> $O/s44_thin.log 2>&1
timeout 300 python tools/prof_product.py 464 66000 66000 10 >> $O/s44_thin.log 2>&1
timeout 300 python tools/prof_product.py 1699 50021 70017 10 >> $O/s44_thin.log 2>&1
timeout 300 python tools/prof_product.py 4464 70000 70000 10 >> $O/s44_thin.log 2>&1
timeout 300 python tools/prof_product.py 65536 65664 128 10 >> $O/s44_thin.log 2>&1
grep shape $O/s44_thin.log
