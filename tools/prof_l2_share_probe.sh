cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/l2_probe
mkdir -p $O
for cfg in "0 0 1000" "0 5 1000" "0 20 1000" "0 100 1000" "0 1000 1000" "1 0 1000" "0 0 6000" "0 20 6000"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum -d $O/$tag -o p -- $R/build/l2_share_probe $cfg > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*results.db" | head -1)
  [ -n "$f" ] && python $R/tools/rocpd_summary.py $f > $O/$tag.summary.txt
  rm -rf $O/$tag
  echo "== mode delay_us chunk_ns = $cfg"; grep -A3 "probe.*dispatches" $O/$tag.summary.txt | grep -v probe
done
