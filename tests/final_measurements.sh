R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/r02_pytest_final.log
python tools/l4_device_timing.py 16384 32768 65536 > $O/r02_l4_device_timing.log 2>&1
python tools/host_api_timing.py > $O/r02_host_api_timing.log 2>&1
python tools/elim_bench.py 65536 > $O/r02_elim_bench.log 2>&1
PYTHONPATH=$R python tools/transpose_trtri_timing.py > $O/r02_transpose_trtri_timing.log 2>&1
ONLY=ple PLE_WHICH=_mzd_ple_russian python tools/l4_device_timing.py 65536 >> $O/r02_l4_device_timing.log 2>&1
for n in 32768 65536; do echo "== LD_PRELOAD=libm4ri_amd.so, n=$n"; M4RI_AMD_STATS=1 LD_PRELOAD=$PWD/m4ri_amd/libm4ri_amd.so timeout 600 oracle/_ref/l4_timing_driver $n 2>&1; done > $O/r02_l4_timing_final.log
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_ple && rocprofv3 --kernel-trace --stats -d /tmp/prof_ple -o t -- python $R/tools/ple_profile_driver.py 65536 > /tmp/prof_ple.log 2>&1; f=$(find /tmp/prof_ple -name "*results.db" | head -1); python $R/tools/rocpd_summary.py $f > $O/r02_ple_65536_trace.summary.txt 2>&1)
bash tools/prof_bench.sh r02_ > $O/prof_bench.log 2>&1
bash tools/prof_transpose.sh r02 > $O/prof_transpose.log 2>&1
python bench.py --steps 20 --warmup 5 > $O/r02_bench_final.json 2> $O/r02_bench_final.err
cat $O/r02_pytest_final.log; cat $O/r02_l4_device_timing.log | grep 65536; tail -4 $O/r02_l4_timing_final.log; tail -c 300 $O/r02_bench_final.json
