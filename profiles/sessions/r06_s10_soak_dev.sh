#!/bin/bash
# round 6, session 2: differential soak of the device-pointer entry points (tests/soak_dev.py), 4 processes sharing the GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r06/soak_dev; mkdir -p $O
SECS=${1:-300}; BASE=${2:-600}
for s in 1 2 3 4; do SOAK_TRACE=${SOAK_TRACE:-} timeout $((SECS + 240)) python tests/soak_dev.py $SECS $((BASE + s)) 5000 > $O/soak_$s.log 2>&1 & done
wait
for s in 1 2 3 4; do grep -v amdgpu.ids $O/soak_$s.log | tail -${TAILN:-4}; done | tee $O/summary.log | cut -c1-700
