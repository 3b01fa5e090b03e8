#!/bin/bash
# the rank-47 table in place: the scheme-pass parity tests, Winograd against scheme on the same box, the bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scheme or four_level or config or sha or 65536" 2>&1 | tail -3
for s in 0 1 0 1; do M4RI_AMD_SCHEME=$s TAG=scheme$s python tools/time_product.py 65536 65536 65536 10 5 2>&1 | grep -v amdgpu.ids; done | tee $O/scheme47_vs_winograd.log
for sh in "32768 32768 32768" "16384 16384 16384" "131072 8192 131072" "65536 65536 16384"; do for s in 0 1; do M4RI_AMD_SCHEME=$s TAG=scheme$s python tools/time_product.py $sh 10 5 2>&1 | grep -v amdgpu.ids; done; done | tee -a $O/scheme47_vs_winograd.log
python bench.py > $O/bench65536_scheme47.json 2> $O/bench65536_scheme47.err; cut -c1-300 $O/bench65536_scheme47.json
