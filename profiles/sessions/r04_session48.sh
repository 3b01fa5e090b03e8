#!/bin/bash
# round 4, GPU session 48: builds of the leaf on ONE box, alternating: A = round's leaf, B = partly filled tiles on the gather-only waves, 3 rows in flight, A dwords 16 rows ahead;
# D / F / E = the same with the A dwords a whole stage (32 rows) ahead and 1 / 2 / 3 rows in flight
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
for rep in 1 2; do
for v in A B D F E; do
  cp build/variants/$v.so m4ri_amd/libm4ri_amd.so
  echo "== variant $v" >> $O/s48_abc.log
  timeout 300 python tools/prof_product.py 65536 65536 65536 8 >> $O/s48_abc.log 2>&1
  timeout 300 python tools/prof_product.py 464 66000 66000 10 >> $O/s48_abc.log 2>&1
  timeout 300 python tools/prof_product.py 2048 65536 65536 10 >> $O/s48_abc.log 2>&1
  timeout 300 python tools/prof_product.py 4464 70000 70000 10 >> $O/s48_abc.log 2>&1
  timeout 300 python tools/prof_product.py 32768 32768 32768 10 >> $O/s48_abc.log 2>&1
done
done
grep "variant\|shape" $O/s48_abc.log | sed 's/pass bytes.*leaf /leaf /'
