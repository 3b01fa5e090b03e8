#!/bin/bash
# round 3, GPU session 10: products in flight (bench.py --inflight 2, the default at N > 1) -- the multi-rank tests again, and the N = 1 line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 3000 python -m pytest tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/r03/s10_pytest_multi.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03/s10_pytest_multi.log
timeout 600 python bench.py --gpus 2 --backend gloo --check --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r03/s10_bench_2ranks_gloo.log 2>&1
timeout 600 python bench.py --gpus 1 --force-dist --variant strassen --steps 5 --warmup 2 --check > gpurun_out/r03/s10_bench_ws1_rccl_strassen.log 2>&1
timeout 600 python bench.py --gpus 1 --force-dist --variant slabs --steps 5 --warmup 2 --check > gpurun_out/r03/s10_bench_ws1_rccl_slabs.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-api > gpurun_out/r03/s10_bench_n1.log 2>&1
tail -6 gpurun_out/r03/s10_pytest_multi.log; for f in 2ranks_gloo ws1_rccl_strassen ws1_rccl_slabs n1; do echo "== $f"; grep -h "OK\|MISMATCH\|Error\|error" gpurun_out/r03/s10_bench_$f.log | head -5; grep -o '"ms_per_step": [0-9.]*\|"latency_ms": [0-9.]*\|"inflight": [0-9]*\|"matches_reference": [a-z]*' gpurun_out/r03/s10_bench_$f.log | tr '\n' ' '; echo; done
