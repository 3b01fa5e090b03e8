#!/bin/bash
# round 3, GPU session 2: HBM ceilings by direction, the passes at a power-of-two and at an off-grid batch stride, the leaf's
# launch time taken apart (build / gather / barrier) with SQ counters, the available counter list
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
rocprofv3 -L > $O/s2_counters_avail.txt 2>&1
python $R/tools/hbm_ceilings.py > $O/s2_hbm_ceilings.log 2>&1
for shape in "65536 65536 65536" "65792 73728 66048"; do
  tag=$(echo $shape | tr ' ' 'x')
  rocprofv3 --kernel-trace --stats -d $O/tr_$tag -o t -- python $R/tools/prof_product.py $shape 5 > $O/s2_trace_$tag.log 2>&1
  python $R/tools/rocpd_summary.py $(find $O/tr_$tag -name "*results.db" | head -1) > $O/s2_trace_$tag.summary.txt 2>&1
  rm -rf $O/tr_$tag
done
for v in base nobuild nogather nobarrier nobuild_nobarrier pd2 pd3 pd2b; do
  $R/build/leaf_check_$v --one 32 1 11 343 > $O/s2_leaf_$v.time.log 2>&1
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
    -d $O/pmc_$v -o p -- $R/build/leaf_check_$v --one 32 1 11 343 > $O/s2_leaf_$v.pmc.log 2>&1
  python $R/tools/rocpd_summary.py $(find $O/pmc_$v -name "*results.db" | head -1) > $O/s2_leaf_$v.pmc.summary.txt 2>&1
  rm -rf $O/pmc_$v
done
# a second counter group on the shipping kernel: what the waves are doing when they are not waiting
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE \
  -d $O/pmc_base2 -o p -- $R/build/leaf_check_base --one 32 1 11 343 > $O/s2_leaf_base.pmc2.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/pmc_base2 -name "*results.db" | head -1) > $O/s2_leaf_base.pmc2.summary.txt 2>&1
rm -rf $O/pmc_base2
cat $O/s2_hbm_ceilings.log; grep -h "^time" $O/s2_leaf_*.time.log; head -12 $O/s2_trace_*.summary.txt
