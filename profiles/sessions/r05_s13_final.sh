#!/bin/bash
# the round's final code: smoke, the driver-style bench line, the rocprofv3 passes over it, the whole -m gpu suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/bench65536_final.json 2> $O/bench65536_final.err; tail -c 300 $O/bench65536_final.err | grep -v amdgpu
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-api --no-traffic > $O/bench65536_final_steps20.json 2>/dev/null
bash tools/prof_bench.sh r05 2>&1 | tail -2
( time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) 2>&1 | tee $O/pytest_gpu_full_final.log
