cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof1
rocprofv3 -L 2>/dev/null | grep -E "SQ_LDS|SQ_INSTS_LDS|SQ_ACTIVE_INST_LDS|SQ_WAIT_INST_LDS|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|GRBM_GUI_ACTIVE|SQ_INSTS_VALU|SQ_ACTIVE_INST_VALU|SQ_WAIT_ANY|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|SQ_INSTS_SALU|SQ_INST_LEVEL_LDS|SQ_WAVES" | head -40 > $R/gpurun_out/prof1/counters_available.txt
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof1/trace -o t -- $R/build/leaf_check --one 32 2 0 64 > $R/gpurun_out/prof1/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $R/gpurun_out/prof1/pmc1 -o p -- $R/build/leaf_check --one 32 2 0 64 > $R/gpurun_out/prof1/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/prof1/pmc2 -o p -- $R/build/leaf_check --one 32 2 0 64 > $R/gpurun_out/prof1/pmc2.log 2>&1
find $R/gpurun_out/prof1 -name "*.csv" | head -20
