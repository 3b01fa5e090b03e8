#!/bin/bash
# round 6, session 2: tests/fuzz_solvers.py (4 processes, the callers one step up: PLE / PLUQ / echelon / TRSM / trtri / transpose / inverse vs the oracle)
# beside a second tests/soak_large.py run (other seed)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r06/soak2; mkdir -p $O
SECS=${1:-360}
for s in 1 2 3 4; do timeout $((SECS + 300)) python tests/fuzz_solvers.py $SECS $((500 + s)) > $O/fuzz_$s.log 2>&1 & done
timeout $((SECS + 300)) python tests/soak_large.py $SECS 402 70000 > $O/large.log 2>&1 &
wait
for s in 1 2 3 4; do grep -v amdgpu.ids $O/fuzz_$s.log | tail -2; done | cut -c1-500
grep -v "^case\|amdgpu.ids" $O/large.log | tail -5 | cut -c1-400
