#!/bin/bash
# usage (on the GPU box): bash tools/prof_pmc_lds.sh <tag> <leaf_check args...>  -> gpurun_out/pmc_<tag>.summary.txt
# LDS counters of one leaf_check run (its own rocprofv3 pass: --pmc only, no tracing)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/pmc_$tag
mkdir -p $O
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/p -o p -- $R/build/leaf_check "$@" > $O/run.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/p -name "*results.db" | head -1) > $R/gpurun_out/pmc_$tag.summary.txt 2>&1
rm -rf $O
