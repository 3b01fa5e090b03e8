#!/bin/bash
# round 4, GPU session 15: the up pass of the four-level schedule with the products meeting in LDS -- parity, timing against the atomic form
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "four_level or three_level" > $O/s15_pytest_fused.log 2>&1
tail -3 $O/s15_pytest_fused.log
for rep in 1 2; do
  timeout 300 python tools/prof_product.py 65536 65536 65536 8 0 3    >> $O/s15_depth_timing.log 2>&1
  M4RI_AMD_UP4=atomic timeout 300 python tools/prof_product.py 65536 65536 65536 8 4096 4 >> $O/s15_depth_timing.log 2>&1
  timeout 300 python tools/prof_product.py 65536 65536 65536 8 4096 4 >> $O/s15_depth_timing.log 2>&1
done
grep shape $O/s15_depth_timing.log
R=$GRAFT_REPO_ROOT
( cd /tmp; rocprofv3 --kernel-trace --stats -d $R/$O/tr15 -o t -- python $R/tools/prof_product.py 65536 65536 65536 5 4096 4 > $R/$O/s15_trace_depth4_fuse4_lds.log 2>&1
  python $R/tools/rocpd_summary.py $(find $R/$O/tr15 -name "*results.db" | head -1) > $R/$O/s15_trace_depth4_fuse4_lds.summary.txt 2>&1; rm -rf $R/$O/tr15 )
grep -i "winograd\|m4rm\|rowwise" $O/s15_trace_depth4_fuse4_lds.summary.txt | head -8
