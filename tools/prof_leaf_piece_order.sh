cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/leaf_perm
mkdir -p $O
echo "checks ok: $($R/build/leaf_check --check-only 2>&1 | grep -c ': ok')  FAIL: $($R/build/leaf_check --check-only 2>&1 | grep -c FAIL)"
for exe in leaf_check leaf_check_skip2; do
  for shape in "8192 8192 8192" "4096 8192 8192"; do
    tag=$exe.$(echo $shape | tr ' ' 'x')
    timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $O/$tag -o p -- $R/build/$exe --traffic $shape 343 0 > $O/$tag.log 2>&1
    f=$(find $O/$tag -name "*results.db" | head -1)
    [ -n "$f" ] && python $R/tools/rocpd_summary.py $f > $O/$tag.summary.txt
    rm -rf $O/$tag
    echo "== $tag"; grep -A2 "m4rm8q_kernel.*dispatches" $O/$tag.summary.txt | grep -v m4rm8q
  done
done
for i in 1 2 3 4; do echo -n "old: "; $R/build/leaf_check_nont0 --traffic 8192 8192 8192 343 0 | grep "^time" | cut -d: -f2; echo -n "new: "; $R/build/leaf_check --traffic 8192 8192 8192 343 0 | grep "^time" | cut -d: -f2; done
