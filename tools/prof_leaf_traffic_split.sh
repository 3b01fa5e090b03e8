#!/bin/bash
# Which operand of the generation-4 leaf is re-fetched?  tools/leaf_check --traffic with every product reading its own
# B and with all products sharing one B (the B traffic vanishes), for 1 / 2 / 4 row tiles per B panel; the fabric read
# requests by size (TCC_EA0_RDREQ, _32B, _64B, _128B) and FETCH_SIZE of the m4rm8q dispatches -> gpurun_out/leaf_split/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/leaf_split
mkdir -p $O
for m in 4096 8192 16384; do
  for share in 0 1; do
    for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
      tag=m$m.share$share.$(echo $set | tr ' ' '+')
      timeout 300 rocprofv3 --pmc $set -d $O/$tag -o p -- $R/build/leaf_check --traffic $m 8192 8192 343 $share > $O/$tag.log 2>&1
      f=$(find $O/$tag -name "*results.db" | head -1)
      [ -n "$f" ] && python $R/tools/rocpd_summary.py $f > $O/$tag.summary.txt
      rm -rf $O/$tag
    done
  done
done
for f in $O/*.summary.txt; do echo "== $(basename $f)"; grep -A4 "m4rm8q_kernel.*dispatches" $f | grep -v m4rm8q; done
