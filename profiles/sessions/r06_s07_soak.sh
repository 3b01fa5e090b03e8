#!/bin/bash
# round 6, session 2: differential soak of the mzd_mul family against the real reference (tests/soak_mul.py), 8 processes sharing the GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r06/soak; mkdir -p $O
SECS=${1:-420}
for s in 1 2 3 4 5 6 7 8; do
  timeout $((SECS + 240)) python tests/soak_mul.py $SECS $((100 + s)) 9000 > $O/soak_$s.log 2>&1 &
done
wait
cat $O/soak_*.log | grep -v amdgpu.ids | tee $O/summary.log | tail -40
