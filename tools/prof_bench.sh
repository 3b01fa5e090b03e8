#!/bin/bash
# rocprofv3 passes over the default bench (run on the GPU box via gpurun); summaries -> gpurun_out/prof_bench/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_bench
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-verify --no-api > $O/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/pmc_lds -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-verify --no-api > $O/pmc_lds.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc_fetch -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-verify --no-api > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $O/pmc_write -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-verify --no-api > $O/pmc_write.log 2>&1
for d in trace pmc_lds pmc_fetch pmc_write; do
  f=$(find $O/$d -name "*results.db" | head -1)
  python $R/tools/rocpd_summary.py $f > $O/$d.summary.txt 2>&1
  rm -rf $O/$d
done
grep -h '"metric"' $O/trace.log | tail -1 > $O/bench_under_trace.json
ls -la $O
python $R/tools/make_leaf_traffic.py $O/pmc_fetch.summary.txt $O/pmc_write.summary.txt 65536 $O/leaf_traffic.json "${1:-}"
