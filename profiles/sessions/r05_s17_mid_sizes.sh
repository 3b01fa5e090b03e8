#!/bin/bash
# resident products at the sizes below the headline: ms per product, against the leaf-bound estimate (units of 4096^3 x 9.72 us / utilisation)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05; mkdir -p $O
for n in 1024 2048 4096 8192 16384 32768; do TAG=n$n python tools/time_product.py $n $n $n 50 10 2>&1 | grep -v amdgpu.ids; done | tee $O/mid_sizes.log
