#!/bin/bash
# the 4 x 4 x 4 scheme passes with the table of Strassen applied twice (R = 49): same leaf count as four Winograd levels, so this measures
# the new pass kernels alone; bit-exactness through the parity tests that reach four fused levels
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05; mkdir -p $O
for round in 1 2; do
  TAG="winograd passes" M4RI_AMD_SCHEME=0 python tools/time_product.py 65536 65536 65536 20 10 2>&1 | grep -v amdgpu.ids
  TAG="scheme passes R=$(grep 'define SCHEME444_R' m4ri_amd/csrc/scheme444.h | awk '{print $3}')" python tools/time_product.py 65536 65536 65536 20 10 2>&1 | grep -v amdgpu.ids
done | tee $O/scheme_vs_winograd.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config3 or four_level or fused or rows_in_blocks or 131072" 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/tools/time_product.py 65536 65536 65536 5 2 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/tr -name "*results.db" | head -1) | tee $GRAFT_REPO_ROOT/$O/scheme_trace.summary.txt | head -12
