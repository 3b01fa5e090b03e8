#!/bin/bash
# round 3, GPU session 4: the whole -m gpu suite + smoke + the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r03/s4_pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03/s4_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03/s4_smoke.log 2>&1
echo "smoke rc $?" >> gpurun_out/r03/s4_smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/s4_bench.json 2> gpurun_out/r03/s4_bench.err
echo "bench rc $?"
tail -5 gpurun_out/r03/s4_pytest_gpu.log; tail -2 gpurun_out/r03/s4_smoke.log; tail -c 2500 gpurun_out/r03/s4_bench.json
