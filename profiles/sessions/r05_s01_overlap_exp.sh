#!/bin/bash
# alternating runs of ONE binary on ONE box: the serial schedule, the passes once more under the leaf, the leaf in 7 launches
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r05
for round in 1 2; do
  TAG="base" python tools/time_product.py 65536 65536 65536 20 10
  TAG="passes-under-leaf (all 3)" M4RI_AMD_OVERLAP_EXP=7 python tools/time_product.py 65536 65536 65536 20 10
  TAG="down4_pack(A) under leaf" M4RI_AMD_OVERLAP_EXP=1 python tools/time_product.py 65536 65536 65536 20 10
  TAG="down4(B) under leaf" M4RI_AMD_OVERLAP_EXP=2 python tools/time_product.py 65536 65536 65536 20 10
  TAG="up4 under leaf" M4RI_AMD_OVERLAP_EXP=4 python tools/time_product.py 65536 65536 65536 20 10
  TAG="leaf in 7 launches" M4RI_AMD_LEAF_GROUPS=7 python tools/time_product.py 65536 65536 65536 20 10
done
