#!/bin/bash
# round 4, GPU session 21: the B-side down pass reading its ancestor once through LDS (M4RI_AMD_DOWN4=lds) -- parity, timing, trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
M4RI_AMD_DOWN4=lds timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "four_level or 65536" > $O/s21_pytest.log 2>&1
tail -3 $O/s21_pytest.log
for rep in 1 2 3; do
  timeout 300 python tools/prof_product.py 65536 65536 65536 8 >> $O/s21_timing.log 2>&1
  M4RI_AMD_DOWN4=lds timeout 300 python tools/prof_product.py 65536 65536 65536 8 >> $O/s21_timing.log 2>&1
done
grep shape $O/s21_timing.log
R=$GRAFT_REPO_ROOT
( cd /tmp; M4RI_AMD_DOWN4=lds rocprofv3 --kernel-trace --stats -d $R/$O/tr21 -o t -- python $R/tools/prof_product.py 65536 65536 65536 5 > $R/$O/s21_trace.log 2>&1
  python $R/tools/rocpd_summary.py $(find $R/$O/tr21 -name "*results.db" | head -1) > $R/$O/s21_trace.summary.txt 2>&1; rm -rf $R/$O/tr21 )
grep -i "winograd\|m4rm" $O/s21_trace.summary.txt | head -6
