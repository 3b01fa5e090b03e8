#!/bin/bash
# round 4, GPU session 12: after the per-device locks of the host entry points -- ThreadSanitizer run, then the whole -m gpu suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4 suppressions=$GRAFT_REPO_ROOT/tools/tsan_suppressions.txt" timeout 900 build/tsan_threads 4 > $O/s12_tsan_threads.log 2>&1
echo "tsan rc $?" >> $O/s12_tsan_threads.log
grep -c "WARNING: ThreadSanitizer" $O/s12_tsan_threads.log; tail -7 $O/s12_tsan_threads.log
timeout 3000 python -m pytest tests -x -q -m gpu > $O/s12_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/s12_pytest_gpu.log
tail -6 $O/s12_pytest_gpu.log
