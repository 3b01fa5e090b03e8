#!/bin/bash
# round 4, GPU session 6: the four-level fused passes -- parity, then depth 3 vs depth 4 (three-level passes + one single level) vs depth 4
# (four-level passes) on one box, kernel traces of the two depth-4 schedules
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "four_level or three_level" > $O/s6_pytest_fused.log 2>&1
tail -5 $O/s6_pytest_fused.log
for rep in 1 2; do
  timeout 300 python tools/prof_product.py 65536 65536 65536 8 0 3    >> $O/s6_depth_timing.log 2>&1
  timeout 300 python tools/prof_product.py 65536 65536 65536 8 4096 3 >> $O/s6_depth_timing.log 2>&1
  timeout 300 python tools/prof_product.py 65536 65536 65536 8 4096 4 >> $O/s6_depth_timing.log 2>&1
done
cat $O/s6_depth_timing.log
cd /tmp
for f in 3 4; do
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/tr_d4_f$f -o t -- python $GRAFT_REPO_ROOT/tools/prof_product.py 65536 65536 65536 5 4096 $f > $GRAFT_REPO_ROOT/$O/s6_trace_depth4_fuse$f.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $GRAFT_REPO_ROOT/$O/tr_d4_f$f -name "*results.db" | head -1) > $GRAFT_REPO_ROOT/$O/s6_trace_depth4_fuse$f.summary.txt 2>&1
  rm -rf $GRAFT_REPO_ROOT/$O/tr_d4_f$f
  grep -i "winograd\|m4rm\|rowwise\|Total" $GRAFT_REPO_ROOT/$O/s6_trace_depth4_fuse$f.summary.txt | head -12
done
