#!/bin/bash
# round 4, GPU session 66: ThreadSanitizer over the host threads driver on the final code (planner, leaf choice, launch model changed since session 12)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4 suppressions=$GRAFT_REPO_ROOT/tools/tsan_suppressions.txt" timeout 900 build/tsan_threads 4 > $O/s66_tsan_threads.log 2>&1
echo "tsan rc $?" >> $O/s66_tsan_threads.log
grep -c "WARNING: ThreadSanitizer" $O/s66_tsan_threads.log; tail -7 $O/s66_tsan_threads.log
timeout 900 python -m pytest tests/test_gpu_threads.py -x -q -m gpu > $O/s66_pytest_threads.log 2>&1; tail -2 $O/s66_pytest_threads.log
