#!/bin/bash
# the rewritten bench.py: N = 1 default line, the 8-rank line on virtual ranks at full size, smoke, and the N > 1 tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r05
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py > gpurun_out/r05/bench65536_default.json 2> gpurun_out/r05/bench65536_default.err ) 2>&1 | grep real
tail -c 600 gpurun_out/r05/bench65536_default.err | grep -v amdgpu.ids
( time python bench.py --gpus 8 --virtual-ranks --watchdog 200 > gpurun_out/r05/bench_peer8_virtual.json 2> gpurun_out/r05/bench_peer8_virtual.err ) 2>&1 | grep real
tail -c 600 gpurun_out/r05/bench_peer8_virtual.err | grep -v amdgpu.ids
timeout 2400 python -m pytest tests/test_gpu_multi.py -x -q 2>&1 | tail -15 | tee gpurun_out/r05/pytest_gpu_multi.log
