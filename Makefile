# Convenience targets; everything they do is plain python / hipcc (see README.md).
PY ?= python

build:            ## hipcc (gfx950) -> m4ri_amd/libm4ri_amd.so, gcc -> oracle/ (+ oracle/_ref when /root/reference exists)
	$(PY) __graft_entry__.py

test: build       ## CPU suite: oracle vs reference vs golden vectors, C ABI symbols, gloo sharding
	$(PY) -m pytest tests -x -q -m "not gpu"

test-gpu: build   ## MI355X suite: parity through the C ABI at BASELINE sizes, residency, drop-in preload
	$(PY) -m pytest tests -x -q -m gpu

bench: build      ## the headline number (one JSON line)
	$(PY) bench.py

leaf-check:       ## developer harness for the leaf kernels (every generation in the build)
	mkdir -p build
	hipcc --offload-arch=gfx950 -O3 -std=c++17 -I m4ri_amd/csrc tools/leaf_check.cpp m4ri_amd/csrc/m4rm_leaf.hip \
	  m4ri_amd/csrc/a4_pack.hip m4ri_amd/csrc/m4rm8q_leaf.hip -o build/leaf_check

.PHONY: build test test-gpu bench leaf-check
