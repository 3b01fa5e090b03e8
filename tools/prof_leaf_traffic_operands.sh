#!/bin/bash
# Fabric read requests of the generation-4 leaf per operand: the normal kernel, one built without its A loads
# (-DK8Q_DEBUG_SKIP=1) and one without its B loads (=2), same launch (343 x 8192^3) -> gpurun_out/leaf_operands/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/leaf_operands
mkdir -p $O
for exe in leaf_check leaf_check_skip1 leaf_check_skip2; do
  for shape in "8192 8192 8192" "4096 8192 8192"; do
    tag=$exe.$(echo $shape | tr ' ' 'x')
    timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum -d $O/$tag -o p -- $R/build/$exe --traffic $shape 343 0 > $O/$tag.log 2>&1
    f=$(find $O/$tag -name "*results.db" | head -1)
    [ -n "$f" ] && python $R/tools/rocpd_summary.py $f > $O/$tag.summary.txt
    rm -rf $O/$tag
    echo "== $tag"; grep -A3 "m4rm8q_kernel.*dispatches" $O/$tag.summary.txt | grep -v m4rm8q
  done
done
