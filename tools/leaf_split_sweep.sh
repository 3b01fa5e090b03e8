#!/bin/bash
# Fewest inner bits per split of a leaf launch (LEAF_MIN_SPLIT_BITS): rebuild engine.o, relink, time short products.
cd $GRAFT_REPO_ROOT
OBJ=m4ri_amd/csrc/_obj
for B in 512 256 128 64; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DLEAF_MIN_SPLIT_BITS=$B -c m4ri_amd/csrc/engine.hip -o $OBJ/engine.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o m4ri_amd/libm4ri_amd.so $OBJ/m4rm_leaf.o $OBJ/a4_pack.o $OBJ/m4rm8q_leaf.o $OBJ/aux_kernels.o $OBJ/engine.o $OBJ/mzd_api.o $OBJ/multi.o $OBJ/trsm.o $OBJ/ple.o $OBJ/elim.o $OBJ/echelon.o $OBJ/solve.o $OBJ/io.o -ldl -lz || exit 1
  echo "== LEAF_MIN_SPLIT_BITS=$B"
  python tools/small_shape_leaf_gens.py 512x512x512 1024x1024x1024 2048x2048x2048 4096x4096x4096 512x512x65536 1024x1024x65536 2048x2048x65536 256x256x65536 6000x6000x6000 8192x8192x8192 16384x16384x16384 2>&1 | grep "^gen"
done
