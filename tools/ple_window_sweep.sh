#!/bin/bash
# Width of the pivot-search window (lanes of the single workgroup of ple_pivots_kernel): rebuild ple.o with
# -DPLE_SLICE_THREADS=T, relink, time PLE on dense and sparse inputs (GPU box; hipcc is in the image).
cd $GRAFT_REPO_ROOT
OBJ=m4ri_amd/csrc/_obj
for T in ${PLE_WINDOWS:-1024 512 256}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DPLE_SLICE_THREADS=$T -c m4ri_amd/csrc/ple.hip -o $OBJ/ple.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o m4ri_amd/libm4ri_amd.so $OBJ/m4rm_leaf.o $OBJ/a4_pack.o $OBJ/m4rm8q_leaf.o $OBJ/aux_kernels.o $OBJ/engine.o $OBJ/mzd_api.o $OBJ/multi.o $OBJ/trsm.o $OBJ/ple.o $OBJ/elim.o $OBJ/echelon.o $OBJ/solve.o $OBJ/io.o -ldl -lz || exit 1
  echo "== window $T lanes"
  python tools/ple_window_timing.py 2>&1 | grep "^ple"
done
