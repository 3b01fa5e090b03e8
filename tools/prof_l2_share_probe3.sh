cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/l2_probe3
mkdir -p $O
for cfg in "1 800 6400 32768 1" "16 800 6400 32768 1" "32 800 6400 32768 1" "16 3000 26000 32768 1"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum -d $O/$tag -o p -- $R/build/l2_share_probe2 $cfg > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*results.db" | head -1)
  [ -n "$f" ] && python $R/tools/rocpd_summary.py $f > $O/$tag.summary.txt
  rm -rf $O/$tag
  echo "== readers gap_ns step_ns chunk_stride = $cfg  (262144 lines)"; grep -A4 "probe.*dispatches" $O/$tag.summary.txt | grep -v probe
done
