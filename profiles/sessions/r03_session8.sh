#!/bin/bash
# round 3, GPU session 8: depth 3 vs 4 again after balancing the passes' grid-stride trips; 131072^3; quick parity of the passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
for cut in 0 4096; do
  rocprofv3 --kernel-trace --stats -d $O/tr_c$cut -o t -- python $R/tools/prof_product.py 65536 65536 65536 5 $cut > $O/s8_trace_cutoff$cut.log 2>&1
  python $R/tools/rocpd_summary.py $(find $O/tr_c$cut -name "*results.db" | head -1) > $O/s8_trace_cutoff$cut.summary.txt 2>&1
  rm -rf $O/tr_c$cut
done
rocprofv3 --kernel-trace --stats -d $O/tr_big -o t -- python $R/tools/prof_product.py 131072 131072 131072 2 > $O/s8_trace_131072.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/tr_big -name "*results.db" | head -1) > $O/s8_trace_131072.summary.txt 2>&1
rm -rf $O/tr_big
cd $R && timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused or config3 or three_level or 131072 or ragged" > $O/s8_pytest.log 2>&1
tail -3 $O/s8_pytest.log
head -12 $O/s8_trace_cutoff0.summary.txt $O/s8_trace_cutoff4096.summary.txt $O/s8_trace_131072.summary.txt
