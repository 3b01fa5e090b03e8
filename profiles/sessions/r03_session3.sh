#!/bin/bash
# round 3, GPU session 3: block-order variants of the three-level down passes; the gather-only leaf variant (reads kept alive)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
for var in 0 1 2 3; do
  M4RI_AMD_PASS_VAR=$var rocprofv3 --kernel-trace --stats -d $O/tr_v$var -o t -- python $R/tools/prof_product.py 65536 65536 65536 6 > $O/s3_trace_passvar$var.log 2>&1
  python $R/tools/rocpd_summary.py $(find $O/tr_v$var -name "*results.db" | head -1) > $O/s3_trace_passvar$var.summary.txt 2>&1
  rm -rf $O/tr_v$var
done
for v in base nobuild nobuild_nobarrier; do
  $R/build/leaf_check_$v --one 32 1 11 343 > $O/s3_leaf_$v.time.log 2>&1
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
    -d $O/pmc_$v -o p -- $R/build/leaf_check_$v --one 32 1 11 343 > $O/s3_leaf_$v.pmc.log 2>&1
  python $R/tools/rocpd_summary.py $(find $O/pmc_$v -name "*results.db" | head -1) > $O/s3_leaf_$v.pmc.summary.txt 2>&1
  rm -rf $O/pmc_$v
done
grep -h "^time" $O/s3_leaf_*.time.log; for var in 0 1 2 3; do echo "== PASS_VAR $var"; grep "winograd" $O/s3_trace_passvar$var.summary.txt; done
