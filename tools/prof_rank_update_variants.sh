#!/bin/bash
# Variants of the PLE's streaming rank update (rows in flight per lane, nontemporal loads / stores): rebuild ple.o with the
# macros, relink, total time of the 1023 launches of a 65536^2 decomposition under rocprofv3 (GPU box).
cd $GRAFT_REPO_ROOT
OBJ=m4ri_amd/csrc/_obj
if [ $# -eq 0 ]; then set -- "-DRU_UNR=4 -DRU_NT=0" "-DRU_UNR=2 -DRU_NT=0" "-DRU_UNR=8 -DRU_NT=0" "-DRU_UNR=4 -DRU_NT=1" "-DRU_UNR=4 -DRU_NT=2" "-DRU_UNR=4 -DRU_NT=3"; fi
for V in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $V -c m4ri_amd/csrc/ple.hip -o $OBJ/ple.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o m4ri_amd/libm4ri_amd.so $OBJ/m4rm_leaf.o $OBJ/a4_pack.o $OBJ/m4rm8q_leaf.o $OBJ/aux_kernels.o $OBJ/engine.o $OBJ/mzd_api.o $OBJ/multi.o $OBJ/trsm.o $OBJ/ple.o $OBJ/elim.o $OBJ/echelon.o $OBJ/solve.o $OBJ/io.o -ldl -lz || exit 1
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_ple && rocprofv3 --kernel-trace --stats -d /tmp/prof_ple -o t -- python $GRAFT_REPO_ROOT/tools/ple_profile_driver.py 65536 > /tmp/prof_ple.log 2>&1
   f=$(find /tmp/prof_ple -name "*results.db" | head -1)
   echo "$V : $(python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $f 2>&1 | grep rank_update | awk '{print "total",$3,"ms avg",$4,"max",$6}')  $(grep ple /tmp/prof_ple.log | tail -1)")
done
