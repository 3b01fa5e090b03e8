"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref/libm4ri_ref.so, built from
/root/reference by oracle/Makefile).  Run in the build container only:

    python tests/golden/make_golden.py [--large]

Two kinds of fixture (the reference tree itself holds none -- all its tests are differential):
  kat_small.npz   full input/output words for small cases of the reference's own test shapes
                  (tests/test_multiplication.c:251-322) plus window cases (tests/test_smallops.c);
  fingerprints.npz  FNV-1a fingerprints (oracle gf2o_fingerprint / Mzd.fingerprint) of large products
                  whose inputs are regenerated from splitmix64 seeds, so nothing big is committed.
Inputs are always Mzd.random(rows, cols, seed) = the fill order of mzd_randomize_custom
(mzd.c:1282-1292) fed with the splitmix64 stream `seed`.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cpu_libs  # noqa: E402
import shapes  # noqa: E402
from m4ri_amd.mzd import Mzd  # noqa: E402

ref = cpu_libs.reference()
assert ref is not None, "build oracle/_ref first: make -C oracle ref"
orc = cpu_libs.oracle()


def kat_small():
    out = {}
    idx = 0
    cases = []
    for (m, l, n, k, cutoff) in shapes.MUL + shapes.EDGE:
        if max(m, l, n) <= 257:
            cases.append(("mul", m, l, n, cutoff))
            cases.append(("m4rm", m, l, n, k))
    for (m, l, n, k, cutoff) in shapes.ADDMUL:
        if max(m, l, n) <= 257:
            cases.append(("addmul", m, l, n, cutoff))
    for (n, k, cutoff) in shapes.SQR:
        if n <= 257:
            cases.append(("sqr", n, n, n, cutoff))
    for (op, m, l, n, par) in cases:
        sa, sb, sc = shapes.seed_of(21, idx, 1), shapes.seed_of(21, idx, 2), shapes.seed_of(21, idx, 3)
        A = Mzd.random(m, l, sa)
        B = A if op == "sqr" else Mzd.random(l, n, sb)
        if op in ("mul", "sqr"):
            C = ref.mul(None, A, B, par)
        elif op == "m4rm":
            if not (m and l and n):
                continue
            C = ref.mul_m4rm(None, A, B, par)
        else:
            C = ref.addmul(Mzd.random(m, n, sc), A, B, par)
        out[f"case{idx}_meta"] = np.array([m, l, n, par], dtype=np.int64)
        out[f"case{idx}_op"] = np.array(op)
        out[f"case{idx}_seeds"] = np.array([sa, sb, sc], dtype=np.uint64)
        out[f"case{idx}_A"] = A.masked()
        out[f"case{idx}_B"] = B.masked()
        out[f"case{idx}_C"] = C.masked()
        idx += 1
    # window cases (cutoff 0, the path test_smallops.c exercises): parent buffers before/after
    for (M, N, m, n) in shapes.SMALLOPS[:2]:
        pat = np.uint64(shapes.SMALLOPS_PATTERN)
        PA, PB, PC = Mzd.init(M, N), Mzd.init(M, N), Mzd.init(M, N)
        for P in (PA, PB, PC):
            P.rows()[:, :] = pat
        k = min(m, n)
        a, b, c = PA.window(0, 0, m, k), PB.window(0, 0, k, n), PC.window(0, 0, m, n)
        a.fill_splitmix(shapes.seed_of(22, M, N, 1))
        b.fill_splitmix(shapes.seed_of(22, M, N, 2))
        c.fill_splitmix(shapes.seed_of(22, M, N, 3))
        ref.addmul(c, a, b, 0)
        out[f"win{M}_{N}_{m}_{n}_parentC"] = PC.buf.copy()
    out["ncases"] = np.array(idx)
    np.savez_compressed(os.path.join(HERE, "kat_small.npz"), **out)
    print("kat_small.npz:", idx, "cases")


def fingerprints(large):
    rows = []
    todo = [("mul", 1025, 1025, 1025, 256), ("mul", 2048, 2048, 4096, 1024), ("mul", 4096, 3528, 4096, 1024),
            ("addmul", 4096, 4096, 4096, 2048), ("mul", 4096, 4096, 4096, 0), ("m4rm", 4096, 4096, 4096, 0),
            ("mul", 1710, 1290, 1000, 256), ("mul", 8192, 8192, 8192, 0), ("mul", 3000, 5000, 7001, 0)]
    if large:
        todo += [("mul", 16384, 16384, 16384, 0)]
    for i, (op, m, l, n, par) in enumerate(todo):
        sa, sb, sc = 1000 + 3 * i, 1001 + 3 * i, 1002 + 3 * i
        t = time.time()
        A, B = Mzd.random(m, l, sa), Mzd.random(l, n, sb)
        if op == "mul":
            C = ref.mul(None, A, B, par)
        elif op == "m4rm":
            C = ref.mul_m4rm(None, A, B, par)
        else:
            C = ref.addmul(Mzd.random(m, n, sc), A, B, par)
        fp = orc.fingerprint(C)
        rows.append((op, m, l, n, par, sa, sb, sc, fp))
        print(op, m, l, n, par, hex(fp), f"{time.time() - t:.1f}s", flush=True)
    np.savez_compressed(os.path.join(HERE, "fingerprints.npz"),
                        ops=np.array([r[0] for r in rows]),
                        meta=np.array([r[1:5] for r in rows], dtype=np.int64),
                        seeds=np.array([r[5:8] for r in rows], dtype=np.uint64),
                        fp=np.array([r[8] for r in rows], dtype=np.uint64))


def fingerprints_xl():
    """BASELINE.json configs 3 and 5 at full size (minutes of reference CPU time, ~5 GiB of RAM)."""
    rows = []
    for (op, m, l, n, par, sa, sb) in [("mul", 65536, 65536, 65536, 0, 3, 4), ("mul", 131072, 8192, 131072, 0, 5, 6)]:
        t = time.time()
        A, B = Mzd.random(m, l, sa), Mzd.random(l, n, sb)
        C = ref.mul(None, A, B, par)
        fp = orc.fingerprint(C)
        rows.append((op, m, l, n, par, sa, sb, 0, fp))
        print(op, m, l, n, par, hex(fp), f"{time.time() - t:.1f}s", flush=True)
        del A, B, C
    np.savez_compressed(os.path.join(HERE, "fingerprints_xl.npz"),
                        ops=np.array([r[0] for r in rows]),
                        meta=np.array([r[1:5] for r in rows], dtype=np.int64),
                        seeds=np.array([r[5:8] for r in rows], dtype=np.uint64),
                        fp=np.array([r[8] for r in rows], dtype=np.uint64))


def fingerprints_xxl(n=131072, sa=7, sb=8, name="fingerprints_xxl.npz"):
    """131072^3 (four Strassen levels on the GPU side): 13 minutes of reference CPU time, ~10 GiB of RAM.
    With --huge: 262144^3 (8 GiB per matrix, ~1.5 h, ~40 GiB of RAM) -> fingerprints_huge.npz."""
    op, m, l, par = "mul", n, n, 0
    t = time.time()
    A, B = Mzd.random(m, l, sa), Mzd.random(l, n, sb)
    C = ref.mul(None, A, B, par)
    fp = orc.fingerprint(C)
    print(op, m, l, n, par, hex(fp), f"{time.time() - t:.1f}s", flush=True)
    np.savez_compressed(os.path.join(HERE, name), ops=np.array([op]),
                        meta=np.array([[m, l, n, par]], dtype=np.int64), seeds=np.array([[sa, sb, 0]], dtype=np.uint64),
                        fp=np.array([fp], dtype=np.uint64))


def fingerprints_ragged_xl():
    """Large shapes off every grid (remainder strips at all three places, partly filled tiles, an inner
    dimension that is not a multiple of 64) -- minutes of reference CPU time each."""
    rows = []
    for (op, m, l, n, par, sa, sb, sc) in [("mul", 100003, 50021, 70017, 0, 21, 22, 0), ("mul", 40000, 131071, 9999, 0, 23, 24, 0),
                                           ("addmul", 65537, 65601, 32897, 0, 25, 26, 27),
                                           ("mul", 32768, 32768, 32768, 1024, 28, 29, 0),   # caller cutoff: 5 levels on the GPU
                                           ("m4rm", 20000, 30011, 10007, 0, 30, 31, 0),     # one leaf launch, no Strassen
                                           ("mul", 8191, 65535, 8193, 0, 32, 33, 0)]:
        t = time.time()
        A, B = Mzd.random(m, l, sa), Mzd.random(l, n, sb)
        C = (ref.mul(None, A, B, par) if op == "mul" else ref.mul_m4rm(None, A, B, par) if op == "m4rm"
             else ref.addmul(Mzd.random(m, n, sc), A, B, par))
        fp = orc.fingerprint(C)
        rows.append((op, m, l, n, par, sa, sb, sc, fp))
        print(op, m, l, n, par, hex(fp), f"{time.time() - t:.1f}s", flush=True)
        del A, B, C
    np.savez_compressed(os.path.join(HERE, "fingerprints_ragged_xl.npz"),
                        ops=np.array([r[0] for r in rows]),
                        meta=np.array([r[1:5] for r in rows], dtype=np.int64),
                        seeds=np.array([r[5:8] for r in rows], dtype=np.uint64),
                        fp=np.array([r[8] for r in rows], dtype=np.uint64))


WINDOW_XL = dict(pa=(30000, 30000, 41), pb=(30000, 30000, 42), pc=(30000, 30000, 43),
                 a=(100, 64, 20100, 16448 + 37), b=(7, 128, 7 + 16384 + 37, 128 + 21000 + 5), c=(9000, 6400, 29000, 6400 + 21000 + 5))


def fingerprint_window_xl():
    """C_window += A_window * B_window on windows of three 30000 x 30000 parents (row offsets anywhere,
    column offsets on word boundaries, widths off the word grid): fingerprint of the WHOLE parent of C.

    The expected value is the reference's product of CLEAN COPIES of the windows, written back into the
    parent under the column mask.  The reference applied to the windows themselves gives different bits in
    the last word column (21005 columns leave a 13-column strip, which goes through the < 54-column
    fallback of _mzd_mul_m4rm, brilliantrussian.c:1063; that path mishandles a windowed B whose excess
    bits are not zero -- DESIGN.md 5).  Both fingerprints are stored; the tests require the first."""
    W = WINDOW_XL
    Pa, Pb, Pc = (Mzd.random(*W[k]) for k in ("pa", "pb", "pc"))
    A, B, C = Pa.window(*W["a"]), Pb.window(*W["b"]), Pc.window(*W["c"])
    t = time.time()
    clean = ref.addmul(C.copy(), A.copy(), B.copy(), 0)
    Pw = Pc.copy()
    ref.addmul(Pw.window(*W["c"]), A, B, 0)       # the reference on the windows themselves
    fp_on_windows = orc.fingerprint(Pw)
    rows, wd = C.rows(), C.width
    rows[:, :wd - 1] = clean.rows()[:, :wd - 1]
    mask = np.uint64((1 << (C.ncols % 64)) - 1) if C.ncols % 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    rows[:, wd - 1] = (rows[:, wd - 1] & ~mask) | (clean.rows()[:, wd - 1] & mask)
    fp = orc.fingerprint(Pc)
    print("window addmul", A.nrows, A.ncols, B.ncols, hex(fp), "(reference on the windows:", hex(fp_on_windows), ")",
          f"{time.time() - t:.1f}s", flush=True)
    np.savez_compressed(os.path.join(HERE, "fingerprint_window_xl.npz"), fp=np.array([fp], dtype=np.uint64),
                        fp_reference_on_windows=np.array([fp_on_windows], dtype=np.uint64))


def sha256_large():
    """SHA-256 (over the valid words, row-major, excess bits masked: Mzd.masked().tobytes()) of the large
    fixtures, beside their 64-bit FNV fingerprints (SURVEY.md 8c asks for SHA-256): the two BASELINE-size
    products, the six ragged XL products and the windowed 30000^2 accumulate.  Every product is recomputed by
    the reference here and its FNV fingerprint compared with the stored one first.  -> sha256.json"""
    import hashlib
    import json
    out = []

    def sha(M):
        return hashlib.sha256(M.masked().tobytes()).hexdigest()

    for fname in ("fingerprints_xl.npz", "fingerprints_ragged_xl.npz"):
        z = np.load(os.path.join(HERE, fname))
        for op, (m, l, n, par), (sa, sb, sc), fp in zip(z["ops"], z["meta"], z["seeds"], z["fp"]):
            op, m, l, n, par, sa, sb, sc = str(op), int(m), int(l), int(n), int(par), int(sa), int(sb), int(sc)
            t = time.time()
            A, B = Mzd.random(m, l, sa), Mzd.random(l, n, sb)
            C = (ref.mul(None, A, B, par) if op == "mul" else ref.mul_m4rm(None, A, B, par) if op == "m4rm"
                 else ref.addmul(Mzd.random(m, n, sc), A, B, par))
            assert orc.fingerprint(C) == int(fp), (fname, op, m, l, n)
            out.append({"op": op, "m": m, "l": l, "n": n, "cutoff": par, "seed_a": sa, "seed_b": sb, "seed_c": sc,
                        "fnv1a": hex(int(fp)), "sha256": sha(C), "source": fname})
            print(op, m, l, n, par, out[-1]["sha256"][:16], f"{time.time() - t:.1f}s", flush=True)
            del A, B, C
            json.dump(out, open(os.path.join(HERE, "sha256.json"), "w"), indent=1)
    W = WINDOW_XL
    Pa, Pb, Pc = (Mzd.random(*W[k]) for k in ("pa", "pb", "pc"))
    A, B, C = Pa.window(*W["a"]), Pb.window(*W["b"]), Pc.window(*W["c"])
    clean = ref.addmul(C.copy(), A.copy(), B.copy(), 0)
    rows, wd = C.rows(), C.width
    rows[:, :wd - 1] = clean.rows()[:, :wd - 1]
    mask = np.uint64((1 << (C.ncols % 64)) - 1) if C.ncols % 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    rows[:, wd - 1] = (rows[:, wd - 1] & ~mask) | (clean.rows()[:, wd - 1] & mask)
    z = np.load(os.path.join(HERE, "fingerprint_window_xl.npz"))
    assert orc.fingerprint(Pc) == int(z["fp"][0])
    out.append({"op": "window_addmul_parent", "m": A.nrows, "l": A.ncols, "n": B.ncols, "cutoff": 0, "seed_a": 41, "seed_b": 42, "seed_c": 43,
                "fnv1a": hex(int(z["fp"][0])), "sha256": sha(Pc), "source": "fingerprint_window_xl.npz"})
    json.dump(out, open(os.path.join(HERE, "sha256.json"), "w"), indent=1)
    print("sha256.json:", len(out), "entries")


PLE_CASES = [("random", 16384, 16384, 51), ("random", 32768, 32768, 52), ("random", 40000, 9000, 53), ("random", 9000, 40000, 54),
             ("lowrank", 20000, 30000, 55), ("zerocols", 12000, 12000, 56)]
TRSM_CASES = [(False, 16384, 20000, 61), (True, 16384, 20000, 62), (False, 30000, 4100, 63), (True, 5000, 70000, 64)]


def ple_input(kind, m, n, seed):
    A = Mzd.random(m, n, seed)
    if kind == "lowrank":
        A = ref.mul(None, Mzd.random(m, 5000, seed + 100), Mzd.random(5000, n, seed + 200), 0)
    elif kind == "zerocols":
        w = A.valid_words()
        w[:, :3] = 0                      # three leading all-zero word columns: the first pivots sit at column 192
        w[:, 40:42] = 0
    elif kind == "defects":               # scattered pivot-free columns, rows that vanish, zero rows: see tests/test_ple_oracle.py
        sys.path.insert(0, os.path.dirname(HERE))
        from test_ple_oracle import _defects
        A = _defects(m, n, seed, m // 16, m // 64)
    return A


def solver_fixtures():
    """mzd_ple and mzd_trsm_{lower,upper}_left of the real reference at sizes where it recurses (ple.c:62-171,
    triangular.c:406-514): SHA-256 over the result's valid words (PLE: followed by P and Q as int32) -> solvers.json"""
    import hashlib
    import json
    out = []
    for kind, m, n, seed in PLE_CASES:
        A = ple_input(kind, m, n, seed)
        t = time.time()
        r, P, Q = ref.ple(A, "mzd_ple")
        h = hashlib.sha256(A.masked().tobytes() + P.astype(np.int32).tobytes() + Q.astype(np.int32).tobytes()).hexdigest()
        out.append({"what": "ple", "kind": kind, "m": m, "n": n, "seed": seed, "rank": r, "sha256": h})
        print("ple", kind, m, n, r, h[:16], f"{time.time() - t:.1f}s", flush=True)
    for upper, mb, nb, seed in TRSM_CASES:
        T, B = Mzd.random(mb, mb, seed), Mzd.random(mb, nb, seed + 1000)
        t = time.time()
        (ref.L.mzd_trsm_upper_left if upper else ref.L.mzd_trsm_lower_left)(T.ptr, B.ptr, 0)
        h = hashlib.sha256(B.masked().tobytes()).hexdigest()
        out.append({"what": "trsm_upper" if upper else "trsm_lower", "m": mb, "n": nb, "seed": seed, "sha256": h})
        print("trsm", upper, mb, nb, h[:16], f"{time.time() - t:.1f}s", flush=True)
    json.dump(out, open(os.path.join(HERE, "solvers.json"), "w"), indent=1)


PLUQ_CASES = [("random", 16384, 16384, 71), ("lowrank", 20000, 30000, 72), ("zerocols", 12000, 12000, 73), ("random", 9000, 40000, 74),
              ("lowrank", 30000, 9000, 75), ("defects", 16384, 16384, 76), ("defects", 12000, 30000, 77), ("defects", 30000, 12000, 78)]


def pluq_fixtures():
    """mzd_pluq of the real reference (ple.c:41-60: PLE, then mzd_apply_p_right_trans_tri): appended to solvers.json."""
    import hashlib
    import json
    path = os.path.join(HERE, "solvers.json")
    out = [e for e in json.load(open(path)) if e["what"] != "pluq" and e.get("kind") != "defects"]
    for kind, m, n, seed in PLUQ_CASES:
        for what in (("pluq", "ple") if kind == "defects" else ("pluq",)):
            A = ple_input(kind, m, n, seed)
            t = time.time()
            r, P, Q = ref.ple(A, "mzd_" + what)
            h = hashlib.sha256(A.masked().tobytes() + P.astype(np.int32).tobytes() + Q.astype(np.int32).tobytes()).hexdigest()
            out.append({"what": what, "kind": kind, "m": m, "n": n, "seed": seed, "rank": r, "sha256": h})
            print(what, kind, m, n, r, h[:16], f"{time.time() - t:.1f}s", flush=True)
    json.dump(out, open(path, "w"), indent=1)


ECHELON_CASES = [(12000, 12000, 81), (8000, 20000, 82), (20000, 8000, 83)]


def echelon_fixtures():
    """mzd_echelonize_pluq and mzd_echelonize_m4ri of the real reference (echelonform.c:29-139), full and not: the two must
    agree; SHA-256 over the result's valid words -> echelon.json"""
    import hashlib
    import json
    out = []
    for m, n, seed in ECHELON_CASES:
        for full in (0, 1):
            hs = []
            for which in ("mzd_echelonize_pluq", "mzd_echelonize_m4ri"):
                A = ple_input("defects", m, n, seed)
                t = time.time()
                r = ref.echelonize(A, full, which)
                hs.append((r, hashlib.sha256(A.masked().tobytes()).hexdigest()))
                print(which, m, n, full, r, hs[-1][1][:16], f"{time.time() - t:.1f}s", flush=True)
            assert hs[0] == hs[1]
            out.append({"m": m, "n": n, "seed": seed, "full": full, "rank": hs[0][0], "sha256": hs[0][1]})
    json.dump(out, open(os.path.join(HERE, "echelon.json"), "w"), indent=1)


TRTRI_CASES = [(8192 + 192, 61), (16384, 62), (20000, 63), (32768, 64)]
TRANSPOSE_CASES = [(20000, 30001, 71), (4099, 65536, 72), (65536, 32768, 73)]


def trtri_transpose_fixtures():
    """mzd_trtri_upper (triangular.c:518-547: every case here takes its halving path) and mzd_transpose (mzd.c:1118-1139) of
    the real reference; SHA-256 over the result's valid words -> trtri_transpose.json.  The inputs are the ones
    tests/test_transpose_trtri_oracle.py's unit_upper(n, seed, keep_lower=True) and Mzd.random(m, n, seed) build."""
    import hashlib
    import json
    sys.path.insert(0, os.path.dirname(HERE))
    from test_transpose_trtri_oracle import unit_upper
    out = []
    for n, seed in TRTRI_CASES:
        U = unit_upper(n, seed, True)
        t = time.time()
        ref.trtri_upper(U)
        h = hashlib.sha256(U.masked().tobytes()).hexdigest()
        print("trtri", n, h[:16], f"{time.time() - t:.1f}s", flush=True)
        out.append({"op": "trtri", "n": n, "seed": seed, "sha256": h})
    for m, n, seed in TRANSPOSE_CASES:
        A = Mzd.random(m, n, seed)
        t = time.time()
        T = ref.transpose(A)
        h = hashlib.sha256(T.masked().tobytes()).hexdigest()
        print("transpose", m, n, h[:16], f"{time.time() - t:.1f}s", flush=True)
        out.append({"op": "transpose", "m": m, "n": n, "seed": seed, "sha256": h})
    json.dump(out, open(os.path.join(HERE, "trtri_transpose.json"), "w"), indent=1)


if __name__ == "__main__":
    if "--trtri" in sys.argv:
        trtri_transpose_fixtures()
    elif "--echelon" in sys.argv:
        echelon_fixtures()
    elif "--pluq" in sys.argv:
        pluq_fixtures()
    elif "--solvers" in sys.argv:
        solver_fixtures()
    elif "--sha" in sys.argv:
        sha256_large()
    elif "--window-xl" in sys.argv:
        fingerprint_window_xl()
    elif "--ragged-xl" in sys.argv:
        fingerprints_ragged_xl()
    elif "--huge" in sys.argv:
        fingerprints_xxl(262144, 9, 10, "fingerprints_huge.npz")
    elif "--xxl" in sys.argv:
        fingerprints_xxl()
    elif "--xl" in sys.argv:
        fingerprints_xl()
    else:
        kat_small()
        fingerprints("--large" in sys.argv)
