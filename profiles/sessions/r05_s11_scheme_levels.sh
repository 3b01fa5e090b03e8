#!/bin/bash
# the scheme passes at 2, 3, 4 (and 5) fused levels: parity tests, then scheme vs Winograd passes by shape (same leaf counts while R = 49)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scheme_passes or config3 or four_level or fused or rows_in_blocks or config5" 2>&1 | tail -5
R=$(grep 'define SCHEME444_R' m4ri_amd/csrc/scheme444.h | awk '{print $3}')
for shape in "65536 65536 65536" "32768 32768 32768" "16384 16384 16384" "131072 8192 131072" "16384 8192 131072"; do
  for round in 1 2; do
    TAG="winograd passes" M4RI_AMD_SCHEME=0 python tools/time_product.py $shape 20 10 2>&1 | grep -v amdgpu.ids
    TAG="scheme passes R=$R" python tools/time_product.py $shape 20 10 2>&1 | grep -v amdgpu.ids
  done
done | tee $O/scheme_vs_winograd_by_shape.log
