#!/bin/bash
# adaptive flip-graph search (plus transitions when stuck) for a rank < 49 scheme of the 4 x 4 x 4 product over GF(2): eight parameter
# variants side by side on the GPU box's 256 host threads (CPU work only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05/flip; mkdir -p $O
T=${1:-1200}; B=build/flipgraph_444_adapt2; IN=tools/flip_ckpt_in.txt
$B 31 $T 47 x $O/ck1.txt $IN 2000000 20000 1 4 > $O/v1.log 2>&1 &
$B 31 $T 47 x $O/ck2.txt $IN 5000000 100000 1 2 > $O/v2.log 2>&1 &
$B 31 $T 47 x $O/ck3.txt $IN 1000000 5000 1 3 > $O/v3.log 2>&1 &
$B 31 $T 47 x $O/ck4.txt $IN 20000000 1000000 1 2 > $O/v4.log 2>&1 &
$B 31 $T 47 s $O/ck5.txt "" 2000000 20000 1 4 > $O/v5.log 2>&1 &
$B 31 $T 47 s $O/ck6.txt "" 5000000 100000 1 2 > $O/v6.log 2>&1 &
$B 31 $T 47 x $O/ck7.txt "" 5000000 50000 1 3 > $O/v7.log 2>&1 &
$B 31 $T 47 x $O/ck8.txt "" 20000000 500000 1 2 > $O/v8.log 2>&1 &
wait
for v in 1 2 3 4 5 6 7 8; do echo "v$v: $(grep '^# ' $O/v$v.log | tail -1 | cut -c1-150)"; done
