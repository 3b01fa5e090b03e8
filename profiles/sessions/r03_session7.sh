#!/bin/bash
# round 3, GPU session 7: row slabs with the all-gather of B under the first product -- bits (gloo, 2 and 4 ranks on one GPU, RCCL at
# world size 1) and the cost of the inner-dimension pieces
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
for n in 2 4; do
  timeout 600 python bench.py --gpus $n --size 16384 --backend gloo --check --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03/s7_slabs_overlap_gloo_$n.log 2>&1
  echo "gloo $n ranks rc $?" >> gpurun_out/r03/s7_slabs_overlap_gloo_$n.log
done
timeout 600 python bench.py --gpus 3 --size 24576 --backend gloo --check --steps 1 --warmup 1 --no-cpu-baseline --slab-overlap 1 > gpurun_out/r03/s7_slabs_overlap_gloo_3.log 2>&1
echo "gloo 3 ranks rc $?" >> gpurun_out/r03/s7_slabs_overlap_gloo_3.log
timeout 600 python bench.py --gpus 1 --force-dist --variant slabs --slab-overlap 1 --size 16384 --steps 2 --warmup 1 --check > gpurun_out/r03/s7_slabs_overlap_rccl_ws1.log 2>&1
echo "rccl ws1 rc $?" >> gpurun_out/r03/s7_slabs_overlap_rccl_ws1.log
timeout 600 python tools/rank_shapes_timing.py > gpurun_out/r03/s7_rank_shapes.log 2>&1
grep -h "rc \|OK\|MISMATCH" gpurun_out/r03/s7_slabs_overlap_*.log | head -30; tail -22 gpurun_out/r03/s7_rank_shapes.log
