#!/bin/bash
# flip-graph walks on the GPU with selective reduction: $1 seconds per setting, then the threshold strings to try, one run each
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05; mkdir -p $O
T=$1; shift
n=0
for thr in "$@"; do
  n=$((n + 1))
  echo "## thresholds $thr"
  timeout $((T + 60)) build/flipgraph_444_gpu $T ${POOL_IN:-none} $O/flip_gpu_sel_pool_$n.txt ${PATHLIM:-5000000} 0 0 16384 200000 ${START:-x} ${SPAN:-4} "$thr" ${LAZY:-0} ${TARGET:-47} > $O/flipgraph_gpu_selective_$n.log 2>&1
  grep -v "^{" $O/flipgraph_gpu_selective_$n.log | grep -v "^#   red" | cut -c1-700 | tail -3
  grep "^#   red" $O/flipgraph_gpu_selective_$n.log | tail -5 | cut -c1-400
  grep -B1 -A50 "^# rank 4[0-8] " $O/flipgraph_gpu_selective_$n.log | head -120
done
