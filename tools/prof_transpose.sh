#!/bin/bash
# The transpose kernel at 65536 x 65536 under rocprofv3 (GPU box): kernel trace, then HBM traffic (FETCH_SIZE and WRITE_SIZE
# in separate passes, MI355X_MICROARCH.md), then where the cycles go.  usage: tools/prof_transpose.sh [tag]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_transpose${1:+_$1}; mkdir -p $O $R/build
hipcc --offload-arch=gfx950 -O3 -o $R/build/transpose_bench $R/tools/transpose_probe.hip || exit 1
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; timeout 200 rocprofv3 "$@" -d $O/$tag -o p -- $R/build/transpose_bench 65536 > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*results.db" | head -1); python $R/tools/rocpd_summary.py $f > $O/$tag.summary.txt 2>&1; grep -A10 "transpose_kernel" $O/$tag.summary.txt | head -12; }
run trace --kernel-trace --stats
run pmc_fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE
run pmc_write --pmc WRITE_SIZE GRBM_GUI_ACTIVE
run pmc_valu --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
run pmc_lds --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
rm -rf $O/*/  # keep the summaries only
