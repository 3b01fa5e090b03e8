#!/bin/bash
# ThreadSanitizer build of the library's host code + the threads driver (developer tool): build/libm4ri_amd_tsan.so, build/tsan_threads
set -e
cd "$(dirname "$0")/.."
mkdir -p build/tsan_obj
SRCS="m4rm_leaf.hip a4_pack.hip m4rm8q_leaf.hip m4rm_small.hip aux_kernels.hip scheme_passes.hip engine.hip mzd_api.hip multi.hip trsm.hip ple.hip elim.hip echelon.hip solve.hip transpose.hip io.cpp small_host.cpp"
for s in $SRCS; do
  hipcc --offload-arch=gfx950 -O1 -g -fsanitize=thread -std=c++17 -fPIC -c m4ri_amd/csrc/$s -o build/tsan_obj/${s%.*}.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=thread -o build/libm4ri_amd_tsan.so build/tsan_obj/*.o -ldl -lz
hipcc -O1 -g -fsanitize=thread -std=c++17 tools/tsan_threads.cpp -o build/tsan_threads -Lbuild -lm4ri_amd_tsan -Wl,-rpath,'$ORIGIN' -lpthread
echo built build/tsan_threads
