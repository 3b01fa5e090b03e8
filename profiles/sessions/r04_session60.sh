#!/bin/bash
# round 4, GPU session 60: the short last round of a leaf launch split by force (M4RI_AMD_TAIL_KSPLIT): does the launch's model leave time on the table at 1.5 rounds?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
for shape in "16384 16384 16384" "32768 32768 32768" "24576 24576 24576" "20480 20480 20480" "65536 65536 65536"; do
  for t in 0 1 2 3 4 5 6 8; do
    echo "== $shape tail_ksplit=$t" >> $O/s60_tail.log
    M4RI_AMD_TAIL_KSPLIT=$t timeout 300 python tools/prof_product.py $shape 30 >> $O/s60_tail.log 2>&1
  done
done
grep "==\|shape" $O/s60_tail.log | sed 's/pass bytes.*leaf /leaf /' | sed 's/, C checksum.*//' | paste - - | sed 's/shape [0-9x]*: levels //' | cut -c1-150
