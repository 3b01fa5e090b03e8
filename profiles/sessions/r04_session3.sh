#!/bin/bash
# round 4, GPU session 3: the multi-GPU schedules behind the C boundary (distributed matrices, m4ri_amd_dmat_mul) on virtual ranks,
# bench.py's transport ladder, the fresh-result path again
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_dmat.py -x -q -m gpu > $O/s3_pytest_dmat.log 2>&1
tail -15 $O/s3_pytest_dmat.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/s3_bench65536_api.json 2> $O/s3_bench65536_api.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04/s3_bench65536_api.json"))
print(d["ms_per_step"], d.get("api"))
PY
timeout 900 python bench.py --gpus 8 --transport peer --virtual-ranks --steps 3 --warmup 1 --no-cpu-baseline > $O/s3_bench_peer8_virtual.json 2> $O/s3_bench_peer8_virtual.err
head -c 1500 $O/s3_bench_peer8_virtual.json; tail -5 $O/s3_bench_peer8_virtual.err
timeout 2400 python -m pytest tests/test_gpu_multi.py -x -q -m gpu > $O/s3_pytest_multi.log 2>&1
tail -15 $O/s3_pytest_multi.log
