#!/bin/bash
# rocprofv3 counter passes over tools/fetch_calibration.py (GPU box); summaries -> gpurun_out/fetch_cal/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fetch_cal
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_RD[A-Z0-9_]*\|TCC_EA_RD[A-Z0-9_]*\|TCC_HIT[A-Z0-9_]*\|TCC_MISS[A-Z0-9_]*\|TCC_REQ[A-Z0-9_]*\|TCC_READ[A-Z0-9_]*" | sort -u > $O/tcc_counters.txt
for shape in tiles_m2 tiles_m1 tiles_m4; do
  for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"; do
    tag=$(echo $set | tr ' ' '+')
    timeout 300 rocprofv3 --pmc $set -d $O/$shape.$tag -o p -- python $R/tools/fetch_calibration.py $shape > $O/$shape.$tag.log 2>&1
    f=$(find $O/$shape.$tag -name "*results.db" | head -1)
    [ -n "$f" ] && python $R/tools/rocpd_summary.py $f > $O/$shape.$tag.summary.txt
    rm -rf $O/$shape.$tag
  done
done
grep -h -A8 "m4rm8q_kernel.*dispatches" $O/*.summary.txt; cat $O/tiles_m*.FETCH_SIZE.log | grep tiles; cat $O/tcc_counters.txt | tr "\n" " "
