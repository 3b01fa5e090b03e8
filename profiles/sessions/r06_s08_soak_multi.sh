#!/bin/bash
# round 6, session 2: differential soak of the multi-device path on virtual ranks against the real reference (tests/soak_multi.py), 4 processes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r06/soak_multi; mkdir -p $O
SECS=${1:-420}; BASE=${2:-200}
for s in 1 2 3 4; do
  SOAK_TRACE=${SOAK_TRACE:-} timeout $((SECS + 240)) python tests/soak_multi.py $SECS $((BASE + s)) 6000 > $O/soak_$s.log 2>&1 &
done
wait
for s in 1 2 3 4; do grep -v amdgpu.ids $O/soak_$s.log | tail -${TAILN:-3}; done | tee $O/summary.log | cut -c1-600
