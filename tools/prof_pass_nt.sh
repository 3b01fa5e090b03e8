cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for nt in 0 1 2 4 7; do
  rm -rf /tmp/prof_nt
  M4RI_AMD_PASS_NT=$nt rocprofv3 --kernel-trace --stats -d /tmp/prof_nt -o t -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-traffic --no-verify > /tmp/nt.log 2>&1
  f=$(find /tmp/prof_nt -name "*results.db" | head -1)
  echo "== PASS_NT=$nt  $(grep -o '"ms_per_step": [0-9.]*' /tmp/nt.log)"
  python $R/tools/rocpd_summary.py $f 2>&1 | grep -E "winograd|m4rm8q" | awk '{print $1, $4}'
done
