#!/bin/bash
# round 4, GPU session 47: three builds of the leaf on ONE box, alternating: A = round's leaf, B = rows of partly filled tiles on the gather-only waves (3 rows in flight),
# C = B + builder-only waves four stages of B rows ahead
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
for rep in 1 2; do
for v in A B C; do
  cp build/variants/$v.so m4ri_amd/libm4ri_amd.so
  echo "== variant $v" >> $O/s47_abc.log
  timeout 300 python tools/prof_product.py 65536 65536 65536 8 >> $O/s47_abc.log 2>&1
  timeout 300 python tools/prof_product.py 464 66000 66000 10 >> $O/s47_abc.log 2>&1
  timeout 300 python tools/prof_product.py 2048 65536 65536 10 >> $O/s47_abc.log 2>&1
  timeout 300 python tools/prof_product.py 4464 70000 70000 10 >> $O/s47_abc.log 2>&1
  timeout 300 python tools/prof_product.py 24576 24576 24576 20 6144 >> $O/s47_abc.log 2>&1
  timeout 300 python tools/prof_product.py 50000 12000 90000 10 >> $O/s47_abc.log 2>&1
done
done
grep "variant\|shape" $O/s47_abc.log
