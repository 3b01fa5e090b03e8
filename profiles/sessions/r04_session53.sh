#!/bin/bash
# round 4, GPU session 53: generation 1 against generation 4 for thin products over the size of B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
for shape in "232 8192 8192" "464 8192 8192" "1000 8192 8192" "232 16384 16384" "464 16384 16384" "1000 16384 16384" "232 33000 33000" "464 33000 33000" "1000 33000 33000" "464 4096 65536" "464 65536 4096" "848 50000 50000" "1232 20000 20000" "1024 1024 65536" "512 65536 512"; do
  for gen in 4 1; do
    echo "== $shape gen=$gen" >> $O/s53_thin_gen.log
    M4RI_AMD_LEAF_GEN=$gen timeout 300 python tools/prof_product.py $shape 20 >> $O/s53_thin_gen.log 2>&1
  done
done
grep "==\|shape" $O/s53_thin_gen.log | sed 's/pass bytes.*leaf /leaf /' | sed 's/, C checksum.*//'
