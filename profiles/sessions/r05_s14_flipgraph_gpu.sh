#!/bin/bash
# the flip-graph walks on the GPU itself (tools/flipgraph_444_gpu.hip): first a sanity descent from the standard algorithm, then the search
#   $1 seconds  $2 pool in (or none)  $3 path limit  $4 plus interval  $5 margin  $6 walks  $7 flips per launch  $8 x = from the standard algorithm  $9 span
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05; mkdir -p $O
timeout 60 build/flipgraph_444_gpu 20 none none 5000000 0 0 8192 100000 x 3 2>&1 | grep -v "^{" | cut -c1-700 | tee $O/flipgraph_gpu_sanity.log
T=${1:-600}
timeout $((T + 60)) build/flipgraph_444_gpu $T ${2:-none} $O/flip_gpu_pool.txt ${3:-5000000} ${4:-0} ${5:-0} ${6:-16384} ${7:-200000} ${8:-x} ${9:-4} > $O/flipgraph_gpu_search.log 2>&1
grep -v "^{" $O/flipgraph_gpu_search.log | cut -c1-900 | tail -22
