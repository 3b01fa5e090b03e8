#!/bin/bash
# round 4, GPU session 62: the leaf launch's split model, A = as it was, B = the tail of a hybrid launch priced once and at 1 bit per slab: many shapes, alternating
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
for rep in 1 2; do
for v in A B; do
  cp build/variants/$v.so m4ri_amd/libm4ri_amd.so
  timeout 900 python tools/many_shapes_timing.py $v >> $O/s62_split_model.log 2>&1
done
done
grep "^[AB] " $O/s62_split_model.log | sort -k2,2 -s | cut -c1-90
