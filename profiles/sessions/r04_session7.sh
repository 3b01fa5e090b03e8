#!/bin/bash
# round 4, GPU session 7: four-level passes with seven waves per workgroup (parity, timing, traces); distributed-matrix tests after the
# creation-clear fix (three times: the failure was a race); the threads driver under ThreadSanitizer with the runtimes suppressed
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "four_level or three_level" > $O/s7_pytest_fused.log 2>&1
tail -3 $O/s7_pytest_fused.log
for rep in 1 2; do
  timeout 300 python tools/prof_product.py 65536 65536 65536 8 0 3    >> $O/s7_depth_timing.log 2>&1
  timeout 300 python tools/prof_product.py 65536 65536 65536 8 4096 4 >> $O/s7_depth_timing.log 2>&1
done
grep shape $O/s7_depth_timing.log
R=$GRAFT_REPO_ROOT
( cd /tmp; rocprofv3 --kernel-trace --stats -d $R/$O/tr7 -o t -- python $R/tools/prof_product.py 65536 65536 65536 5 4096 4 > $R/$O/s7_trace_depth4_fuse4.log 2>&1
  python $R/tools/rocpd_summary.py $(find $R/$O/tr7 -name "*results.db" | head -1) > $R/$O/s7_trace_depth4_fuse4.summary.txt 2>&1; rm -rf $R/$O/tr7 )
grep -i "winograd\|m4rm\|rowwise" $O/s7_trace_depth4_fuse4.summary.txt | head -8
for k in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_dmat.py -x -q -m gpu 2>&1 | tail -2; done > $O/s7_pytest_dmat_x3.log 2>&1
cat $O/s7_pytest_dmat_x3.log
TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4 suppressions=$R/tools/tsan_suppressions.txt" timeout 900 build/tsan_threads 3 > $O/s7_tsan_threads.log 2>&1
echo "tsan rc $?" >> $O/s7_tsan_threads.log
grep -c "WARNING: ThreadSanitizer" $O/s7_tsan_threads.log; tail -7 $O/s7_tsan_threads.log
