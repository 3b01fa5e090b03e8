"""Build libm4ri_amd.so (HIP kernels + engine + C ABI) for gfx950, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the snapshot.  `python -m m4ri_amd.build` or
`__graft_entry__.build()`.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libm4ri_amd.so")
SOURCES = ["m4rm_leaf.hip", "a4_pack.hip", "m4rm8q_leaf.hip", "m4rm_small.hip", "aux_kernels.hip", "scheme_passes.hip", "engine.hip", "mzd_api.hip", "multi.hip", "trsm.hip", "ple.hip", "elim.hip", "echelon.hip", "solve.hip", "transpose.hip", "io.cpp", "small_host.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fvisibility-inlines-hidden"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libm4ri_amd.so cannot be built (no CPU fallback exists)")
    return exe


def declared_functions() -> list[str]:
    """The functions include/m4ri_amd.h declares: M4RI's own names for this path and the m4ri_amd_* extensions.
    They are the library's whole dynamic symbol table (export_map)."""
    import re
    text = open(os.path.join(HERE, "..", "include", "m4ri_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^[A-Za-z_][\w\s\*]*?\b(\w+)\s*\([^;{]*\)\s*;", text, flags=re.M)
    return sorted(set(n for n in names if n not in ("defined",)))


def export_map() -> str:
    """Linker version script with the explicit export list.  The library is an LD_PRELOAD interposer: every stray
    global (the engine's gf2_* launchers, libstdc++ template instantiations) would be a collision risk in the host
    program, so only the declared names leave the .so; everything else is local."""
    path = os.path.join(OBJ, "exports.map")
    text = "{\n  global:\n" + "".join(f"    {n};\n" for n in declared_functions()) + "  local:\n    *;\n};\n"
    if not os.path.exists(path) or open(path).read() != text:
        with open(path, "w") as f:
            f.write(text)
    return path


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, "gf2_common.h"), os.path.join(CSRC, "scheme444.h"), os.path.join(HERE, "..", "include", "m4ri_amd.h")]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *FLAGS, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[m4ri_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    emap = export_map()
    if force or jobs or _stale(LIB, objs + [emap]):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, f"-Wl,--version-script={emap}", "-ldl", "-lz"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
