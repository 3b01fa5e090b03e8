#!/bin/bash
# round 4, GPU session 68: nontemporal stores in the four-level passes as they are now (LDS forms on all three): every combination, alternating twice
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
for rep in 1 2; do
for nt in 0 1 2 3 4 5 6 7; do
  echo "== PASS_NT=$nt" >> $O/s68_nt.log
  M4RI_AMD_PASS_NT=$nt timeout 300 python tools/prof_product.py 65536 65536 65536 10 >> $O/s68_nt.log 2>&1
done
done
grep "==\|shape" $O/s68_nt.log | sed 's/pass bytes.*leaf /leaf /' | sed 's/, C checksum.*//' | paste - - | sed 's/shape [0-9x]*: levels 4, leaf 4096x4096x4096 x2401, //'
