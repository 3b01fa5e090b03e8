#!/bin/bash
# the whole -m gpu suite on the round's code, ThreadSanitizer over the host threads (lanes and cross-thread pin queries included), the
# rocprofv3 passes over the default bench; a depth sweep when the box is of the slow class
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05; mkdir -p $O
TAG=probe python tools/time_product.py 65536 65536 65536 10 5 2>&1 | grep -v amdgpu.ids | tee $O/box_speed.log
TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4 suppressions=$PWD/tools/tsan_suppressions.txt" timeout 900 build/tsan_threads 4 > $O/tsan_host_threads.log 2>&1
echo "tsan rc $?" >> $O/tsan_host_threads.log
echo "ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' $O/tsan_host_threads.log)" | tee -a $O/tsan_host_threads.log; tail -9 $O/tsan_host_threads.log
( time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) 2>&1 | tee $O/pytest_gpu_full.log
bash tools/prof_bench.sh r05 2>&1 | tail -3
ms=$(awk '{for(i=1;i<=NF;i++) if ($i=="ms/product,") print $(i-1)}' $O/box_speed.log | head -1)
if python -c "import sys; sys.exit(0 if float('$ms') > 28.4 else 1)"; then
  python tools/depth_model_sweep.py 65536,65536,65536 32768,32768,32768 16384,16384,16384 131072,8192,131072 16384,8192,131072 2>&1 | grep -v amdgpu.ids | tee $O/depth_model_slow_box.log
fi
