#!/bin/bash
# round 6, session 3: differential soaks after the small leaf and the second host routine -- soak_mul (host entry points, dims <= 3000 so that most
# products are small ones) and soak_dev (device-pointer entry points, batches), two processes each sharing the GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r06s3/soak_small; mkdir -p $O
SECS=${1:-600}; BASE=${2:-800}
timeout $((SECS + 240)) python tests/soak_mul.py $SECS $((BASE + 1)) 3000 > $O/soak_mul_1.log 2>&1 &
timeout $((SECS + 240)) python tests/soak_mul.py $SECS $((BASE + 2)) 9000 > $O/soak_mul_2.log 2>&1 &
timeout $((SECS + 240)) python tests/soak_dev.py $SECS $((BASE + 3)) 3000 > $O/soak_dev_3.log 2>&1 &
timeout $((SECS + 240)) python tests/soak_dev.py $SECS $((BASE + 4)) 5000 > $O/soak_dev_4.log 2>&1 &
wait
for f in soak_mul_1 soak_mul_2 soak_dev_3 soak_dev_4; do grep -v amdgpu.ids $O/$f.log | tail -2; done | tee $O/summary.log | cut -c1-400
