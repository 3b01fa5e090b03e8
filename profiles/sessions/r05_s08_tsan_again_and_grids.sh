#!/bin/bash
# ThreadSanitizer again after the PinLock fix (its destructor read the pin's device after unpin had erased the entry), the pin / thread tests, then
# the block grids / orders of the host pipeline
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05; mkdir -p $O
TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4 suppressions=$PWD/tools/tsan_suppressions.txt" timeout 900 build/tsan_threads 4 > $O/tsan_host_threads.log 2>&1
echo "tsan rc $?" >> $O/tsan_host_threads.log
echo "ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' $O/tsan_host_threads.log)" | tee -a $O/tsan_host_threads.log; tail -9 $O/tsan_host_threads.log
timeout 900 python -m pytest tests/test_gpu_residency.py tests/test_gpu_threads.py -x -q -m gpu 2>&1 | tail -3
bash profiles/sessions/r05_s07_pipe_grids.sh
