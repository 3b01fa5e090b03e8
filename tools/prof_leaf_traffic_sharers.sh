cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/leaf_split2
mkdir -p $O
for n in ${SHARER_NS:-512 1024 2048 4096 8192}; do
  for share in 1 0; do
    tag=n$n.share$share
    timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum -d $O/$tag -o p -- $R/build/leaf_check --traffic 8192 8192 $n 343 $share > $O/$tag.log 2>&1
    f=$(find $O/$tag -name "*results.db" | head -1)
    [ -n "$f" ] && python $R/tools/rocpd_summary.py $f > $O/$tag.summary.txt
    rm -rf $O/$tag
    echo "== $tag"; grep -A4 "m4rm8q_kernel.*dispatches" $O/$tag.summary.txt | grep -v m4rm8q; grep "m4rm8q" $O/$tag.summary.txt | head -1
  done
done
