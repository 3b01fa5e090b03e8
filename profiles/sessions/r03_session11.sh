#!/bin/bash
# round 3, GPU session 11: odd world sizes / shapes through bench.py's multi-rank paths (gloo on one GPU, --check on every rank)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
O=gpurun_out/r03/s11_sweep.log
: > $O
run() { echo "=== bench.py $*" >> $O; timeout 900 python bench.py "$@" --backend gloo --check --steps 3 --warmup 1 --no-cpu-baseline >> $O 2>&1; echo "rc $?" >> $O; }
run --gpus 5 --size 65536
run --gpus 7 --size 32768
run --gpus 3 --size 24576 --variant strassen
run --gpus 6 --size 49152 --variant strassen --overlap 3x2
run --gpus 4 --dims 70000,66000,65536
run --gpus 2 --dims 50000,40000,30001
run --gpus 8 --dims 20000,131072,8192
run --gpus 4 --size 32768 --inflight 1
run --gpus 4 --size 32768 --variant strassen --inflight 2 --shard-levels 1
grep -c -- "-> OK" $O; grep -c "MISMATCH\|Traceback" $O; grep "^rc\|=== " $O | paste - - | head -20
grep -o '"variant": "[a-z]*"\|"inflight": [0-9]\|"overlap_chunks": \[[0-9, ]*\]\|"all_gather_under_first_product": [a-z]*' $O | paste - - - | head -20
