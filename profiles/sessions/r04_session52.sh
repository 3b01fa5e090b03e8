#!/bin/bash
# round 4, GPU session 52: which leaf generation for thin products (few rows, long inner dimension, many columns)?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
for m in 128 192 256 464 768 1024 1699 2048; do
  for gen in 4 1; do
    echo "== m=$m gen=$gen" >> $O/s52_thin_gen.log
    M4RI_AMD_LEAF_GEN=$gen timeout 300 python tools/prof_product.py $m 66000 66000 10 >> $O/s52_thin_gen.log 2>&1
  done
done
grep "==\|shape" $O/s52_thin_gen.log | sed 's/pass bytes.*leaf /leaf /'
