#!/bin/bash
# round 4, GPU session 18: staggered first round of the leaf launch (M4RI_AMD_LEAF_STAGGER = percent of a tile period)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
  for st in 0 50 100 200; do
    echo "stagger $st" >> $O/s18_stagger.log
    M4RI_AMD_LEAF_STAGGER=$st timeout 300 python tools/prof_product.py 65536 65536 65536 8 >> $O/s18_stagger.log 2>&1
    M4RI_AMD_LEAF_STAGGER=$st timeout 300 python tools/prof_product.py 65536 65536 65536 8 8192 3 >> $O/s18_stagger.log 2>&1
  done
done
grep -v amdgpu.ids $O/s18_stagger.log
