#!/bin/bash
# round 4, GPU session 23: the A-side four-level pack pass without the transpose (M4RI_AMD_DOWN4_PACK=lds) -- parity, timing, trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
M4RI_AMD_DOWN4_PACK=lds timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "four_level or 65536" > $O/s23_pytest.log 2>&1
tail -3 $O/s23_pytest.log
for rep in 1 2 3; do
  timeout 300 python tools/prof_product.py 65536 65536 65536 8 >> $O/s23_timing.log 2>&1
  M4RI_AMD_DOWN4_PACK=lds timeout 300 python tools/prof_product.py 65536 65536 65536 8 >> $O/s23_timing.log 2>&1
done
grep shape $O/s23_timing.log
R=$GRAFT_REPO_ROOT
( cd /tmp; M4RI_AMD_DOWN4_PACK=lds rocprofv3 --kernel-trace --stats -d $R/$O/tr23 -o t -- python $R/tools/prof_product.py 65536 65536 65536 5 > $R/$O/s23_trace.log 2>&1
  python $R/tools/rocpd_summary.py $(find $R/$O/tr23 -name "*results.db" | head -1) > $R/$O/s23_trace.summary.txt 2>&1; rm -rf $R/$O/tr23 )
grep -i "winograd\|m4rm" $O/s23_trace.summary.txt | head -6
