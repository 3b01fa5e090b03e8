#!/bin/bash
# round 3, GPU session 6: Strassen depth 4 at 65536 taken apart (leaf vs passes), same box as depth 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
for cut in 0 4096; do
  rocprofv3 --kernel-trace --stats -d $O/tr_c$cut -o t -- python $R/tools/prof_product.py 65536 65536 65536 5 $cut > $O/s6_trace_cutoff$cut.log 2>&1
  python $R/tools/rocpd_summary.py $(find $O/tr_c$cut -name "*results.db" | head -1) > $O/s6_trace_cutoff$cut.summary.txt 2>&1
  rm -rf $O/tr_c$cut
done
tail -2 $O/s6_trace_cutoff*.log; head -12 $O/s6_trace_cutoff*.summary.txt
