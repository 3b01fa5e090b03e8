#!/bin/bash
# round 4, GPU session 5: distributed-matrix and thread tests again, the threads driver under ThreadSanitizer, then the whole -m gpu suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_dmat.py tests/test_gpu_threads.py -x -q -m gpu > $O/s5_pytest_dmat_threads.log 2>&1
tail -8 $O/s5_pytest_dmat_threads.log
TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4" timeout 900 build/tsan_threads 3 > $O/s5_tsan_threads.log 2>&1
echo "tsan rc $?" >> $O/s5_tsan_threads.log
grep -c "WARNING: ThreadSanitizer" $O/s5_tsan_threads.log; tail -5 $O/s5_tsan_threads.log
timeout 3000 python -m pytest tests -x -q -m gpu > $O/s5_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/s5_pytest_gpu.log
tail -8 $O/s5_pytest_gpu.log
