R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for rows in 128 256 512 1024; do for tw in 32; do
  rm -rf /tmp/prof_ple
  M4RI_AMD_RU_ROWS=$rows M4RI_AMD_RU_TW=$tw rocprofv3 --kernel-trace --stats -d /tmp/prof_ple -o t -- python $R/tools/ple_profile_driver.py 65536 > /tmp/prof_ple.log 2>&1
  f=$(find /tmp/prof_ple -name "*results.db" | head -1)
  echo "rows=$rows tw=$tw $(python $R/tools/rocpd_summary.py $f 2>&1 | grep rank_update | awk '{print "total",$3,"avg",$4,"max",$6}')"
done; done
