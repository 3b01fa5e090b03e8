#!/bin/bash
# Rows of the TRSM's inverted diagonal blocks: rebuild trsm.o with -DTRSM_TB=T, relink, time resident solves (GPU box).
cd $GRAFT_REPO_ROOT
OBJ=m4ri_amd/csrc/_obj
for T in 512 1024 256; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DTRSM_TB=$T -c m4ri_amd/csrc/trsm.hip -o $OBJ/trsm.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o m4ri_amd/libm4ri_amd.so $OBJ/m4rm_leaf.o $OBJ/a4_pack.o $OBJ/m4rm8q_leaf.o $OBJ/aux_kernels.o $OBJ/engine.o $OBJ/mzd_api.o $OBJ/multi.o $OBJ/trsm.o $OBJ/ple.o $OBJ/elim.o $OBJ/echelon.o $OBJ/solve.o $OBJ/io.o -ldl -lz || exit 1
  echo "== blocks of $T rows"
  python tools/l4_device_timing.py 2>&1 | grep "trsm.*resident"
done
