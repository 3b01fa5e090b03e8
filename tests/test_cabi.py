"""CPU-side checks of the drop-in boundary: libm4ri_amd.so builds for gfx950, loads without a GPU,
exports every function include/m4ri_amd.h declares, and shares M4RI's 64-byte descriptor layout.
No compute calls here (there is no GPU in the build container)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

import m4ri_amd
from m4ri_amd.mzd import Mzd, MzdStruct

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "m4ri_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^[A-Za-z_][\w\s\*]*?\b(\w+)\s*\([^;{]*\)\s*;", text, flags=re.M)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_library_builds_and_loads():
    from m4ri_amd import build
    path = build.build(verbose=False)
    assert os.path.exists(path)
    assert m4ri_amd.lib() is not None


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 24, names
    L = m4ri_amd.lib()
    for n in names:
        assert hasattr(L, n), f"include/m4ri_amd.h declares {n} but libm4ri_amd.so does not export it"
        assert n in m4ri_amd.SYMBOLS, f"{n} has no ctypes signature in m4ri_amd.SYMBOLS"
    # and nothing bound in Python is missing from the header
    for n in m4ri_amd.SYMBOLS:
        assert n in names, f"{n} is bound but not declared in include/m4ri_amd.h"


def test_nothing_but_the_declared_names_is_exported():
    """The library is an LD_PRELOAD interposer: every stray global is a collision risk in the host program.  Its dynamic
    symbol table is exactly the header's list (m4ri_amd/build.py: export_map) -- no gf2_* launchers, no libstdc++
    template instantiations."""
    out = subprocess.run(["nm", "-D", "--defined-only", m4ri_amd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert exported == set(declared_functions()), sorted(exported ^ set(declared_functions()))


def test_m4ri_drop_in_names_present():
    # the symbols a replacement for this path must export
    need = ["mzd_mul", "mzd_addmul", "_mzd_mul_even", "_mzd_addmul_even", "_mzd_addmul", "mzd_mul_m4rm",
            "mzd_addmul_m4rm", "_mzd_mul_m4rm", "_mzd_sqr_even", "_mzd_addsqr_even", "mzd_mul_mp", "mzd_addmul_mp"]
    out = subprocess.run(["nm", "-D", "--defined-only", m4ri_amd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert set(need) <= exported
    # the library must NOT define the allocator names: C == NULL results have to come from the host
    # program's libm4ri (dlsym), or callers could not mzd_free() them
    assert "mzd_init" not in exported and "mzd_free" not in exported


def test_descriptor_layout_matches_mzd_t():
    # offsets probed on the reference build: nrows@0 ncols@4 width@8 rowstride@16 flags@24
    # high_bitmask@48 data@56, sizeof 64
    assert ctypes.sizeof(MzdStruct) == 64
    for field, off in [("nrows", 0), ("ncols", 4), ("width", 8), ("rowstride", 16), ("flags", 24), ("high_bitmask", 48), ("data", 56)]:
        assert getattr(MzdStruct, field).offset == off
    m = Mzd.init(5, 70)
    assert (m.width, m.rowstride, m.high_bitmask, m.struct.flags) == (2, 2, (1 << 6) - 1, 0x2)
    w = m.window(1, 64, 4, 70)
    assert (w.nrows, w.ncols, w.width, w.rowstride, w.struct.flags) == (3, 6, 1, 2, 0x6)


def test_host_allocator_contract(reference=None):
    L = m4ri_amd.lib()
    p = L.m4ri_amd_mzd_init(3, 130)
    s = p.contents
    assert (s.nrows, s.ncols, s.width, s.rowstride, s.high_bitmask) == (3, 130, 3, 4, 3)
    assert all(s.data[i] == 0 for i in range(12))
    L.m4ri_amd_mzd_free(p)


def test_missing_library_fails_loudly(tmp_path):
    code = ("import m4ri_amd, sys; m4ri_amd.LIB_PATH = '/nonexistent/libm4ri_amd.so'\n"
            "try:\n    m4ri_amd.lib()\nexcept RuntimeError as e:\n    print('LOUD', e); sys.exit(0)\nsys.exit(1)")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0 and "no CPU fallback" in r.stdout


def test_product_path_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under m4ri_amd/ may reference it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "m4ri_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                t = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"gf2_oracle|liboracle|oracle/|cpu_libs|libm4ri_ref", t):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_header_is_plain_c99(tmp_path):
    """include/m4ri_amd.h is the contract a C program (or M4RI itself) compiles against: it must be
    valid C99 on its own, and -- with M4RI_AMD_NO_MZD_T -- next to M4RI's own mzd_t."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    inc = os.path.join(ROOT, "include")
    src = tmp_path / "uses_header.c"
    src.write_text('#include <m4ri_amd.h>\n'
                   'int main(void) { mzd_t *A = m4ri_amd_mzd_init(8, 8); mzd_t *C = mzd_mul(0, A, A, 0);\n'
                   '  m4ri_amd_pin(A); m4ri_amd_unpin(A); m4ri_amd_mzd_free(A); m4ri_amd_mzd_free(C);\n'
                   '  return m4ri_amd_plan_levels(65536, 65536, 65536, 0) != 3; }\n')
    r = subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ref_inc = "/root/reference"
    if os.path.exists(os.path.join(ref_inc, "m4ri", "mzd.h")) and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "inc_seq")):
        src2 = tmp_path / "next_to_m4ri.c"
        src2.write_text('#include <m4ri/m4ri.h>\n#define M4RI_AMD_NO_MZD_T\n#include <m4ri_amd.h>\n'
                        'int main(void) { return sizeof(mzd_t) != 64; }\n')
        cfg = os.path.join(ROOT, "oracle", "_ref", "inc_seq")
        r = subprocess.run([gcc, "-std=gnu99", "-Wall", "-fsyntax-only", "-I", inc, "-I", cfg, "-I", os.path.join(cfg, "m4ri"),
                            "-I", ref_inc, str(src2)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        # the binding hunk INTEGRATION.md shows a maintainer, as a compilable unit
        r = subprocess.run([gcc, "-std=gnu99", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, "-I", cfg, "-I", os.path.join(cfg, "m4ri"),
                            "-I", ref_inc, os.path.join(ROOT, "tests", "integration_stub.c")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
