#!/bin/bash
# round 4, GPU session 13: the single-transfer path of small GPU products -- parity (the whole parity file + thread / small tests), crossover again
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_threads.py tests/test_small_products.py tests/test_gpu_residency.py tests/test_gpu_host_pipeline.py tests/test_dropin_preload.py tests/test_gpu_add.py -x -q -m gpu > $O/s13_pytest.log 2>&1
tail -4 $O/s13_pytest.log
timeout 900 python tests/crossover_cpu_gpu.py > $O/s13_crossover_cpu_gpu.log 2>&1
head -26 $O/s13_crossover_cpu_gpu.log
