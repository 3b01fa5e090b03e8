cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/leaf_nt
mkdir -p $O
for exe in leaf_check_nont0 leaf_check; do
  for set in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$exe.$(echo $set | tr ' ' '+')
    timeout 300 rocprofv3 --pmc $set -d $O/$tag -o p -- $R/build/$exe --traffic 8192 8192 8192 343 0 > $O/$tag.log 2>&1
    f=$(find $O/$tag -name "*results.db" | head -1)
    [ -n "$f" ] && python $R/tools/rocpd_summary.py $f > $O/$tag.summary.txt
    rm -rf $O/$tag
    echo "== $tag"; grep -A3 "m4rm8q_kernel.*dispatches" $O/$tag.summary.txt | grep -v m4rm8q; grep "^time" $O/$tag.log
  done
done
for i in 1 2 3; do $R/build/leaf_check_nont0 --traffic 8192 8192 8192 343 0 | grep "^time"; $R/build/leaf_check --traffic 8192 8192 8192 343 0 | grep "^time"; done
