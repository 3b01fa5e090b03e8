#!/bin/bash
# the flip-graph search for a rank-47 scheme of the 4 x 4 x 4 product over GF(2) on the GPU box's 256 host threads (CPU work only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05; mkdir -p $O
IN=""; [ -f tools/flip_ckpt_in.txt ] && IN=tools/flip_ckpt_in.txt
nproc
build/flipgraph_444_new 250 ${1:-1500} 47 x $O/flip_ckpt.txt "$IN" 60000000 > $O/flipgraph.log 2>&1
grep "^# " $O/flipgraph.log | tail -12
