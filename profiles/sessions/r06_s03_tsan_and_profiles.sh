#!/bin/bash
# round 6: ThreadSanitizer over the host threads with the ranks' sub-products grouped into batched products (and without), then the rocprofv3
# passes over the default bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r06; mkdir -p $O
for g in 1 3; do
  TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4 suppressions=$PWD/tools/tsan_suppressions.txt" timeout 900 build/tsan_threads 4 $g > $O/tsan_host_threads_group$g.log 2>&1
  echo "tsan rc $?" >> $O/tsan_host_threads_group$g.log
  echo "group $g: ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' $O/tsan_host_threads_group$g.log)" | tee -a $O/tsan_host_threads_group$g.log; tail -6 $O/tsan_host_threads_group$g.log
done
bash tools/prof_bench.sh r06 2>&1 | tail -3
cp -r gpurun_out/prof_bench $O/prof_bench_mid_round
