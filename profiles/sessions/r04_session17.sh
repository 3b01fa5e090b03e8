#!/bin/bash
# round 4, GPU session 17: the new default depth (leaves of 4096 inner bits, four-level passes) -- parity files, the sweep again with
# the pack kernels' one-barrier-per-seven-outputs, kernel trace of the default product, bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dmat.py tests/test_gpu_residency.py tests/test_gpu_host_pipeline.py tests/test_gpu_trsm.py tests/test_gpu_ple.py -x -q -m gpu > $O/s17_pytest.log 2>&1
tail -4 $O/s17_pytest.log
timeout 1500 python tools/depth_rule_sweep.py > $O/s17_depth_rule_sweep.log 2>&1
grep -v amdgpu.ids $O/s17_depth_rule_sweep.log
R=$GRAFT_REPO_ROOT
( cd /tmp; rocprofv3 --kernel-trace --stats -d $R/$O/tr17 -o t -- python $R/tools/prof_product.py 65536 65536 65536 5 > $R/$O/s17_trace_default.log 2>&1
  python $R/tools/rocpd_summary.py $(find $R/$O/tr17 -name "*results.db" | head -1) > $R/$O/s17_trace_default.summary.txt 2>&1; rm -rf $R/$O/tr17 )
grep -i "winograd\|m4rm\|rowwise" $O/s17_trace_default.summary.txt | head -8
( cd /tmp; rocprofv3 --kernel-trace --stats -d $R/$O/tr17b -o t -- python $R/tools/prof_product.py 65536 65536 65536 5 8192 3 > $R/$O/s17_trace_depth3.log 2>&1
  python $R/tools/rocpd_summary.py $(find $R/$O/tr17b -name "*results.db" | head -1) > $R/$O/s17_trace_depth3.summary.txt 2>&1; rm -rf $R/$O/tr17b )
grep -i "winograd\|m4rm\|rowwise" $O/s17_trace_depth3.summary.txt | head -8
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/s17_bench.json 2> $O/s17_bench.err
head -c 500 $O/s17_bench.json; echo
