#!/usr/bin/env python3
"""Differential soak of the hot path (a script, not a pytest module; it lives under tests/ because it runs the checker; GPU box):
random shapes, windows, strides, cutoffs, fuse depths and entry points of the mzd_mul family through libm4ri_amd.so's C ABI against
the REAL reference built into oracle/_ref (the CPU oracle when that build is absent), bit for bit, for a wall-clock budget.

    python tests/soak_mul.py [seconds] [seed] [max_dim]

Every case draws: an entry point (mzd_mul with C == NULL or a dirty C, mzd_addmul, _mzd_mul_even, _mzd_addmul_even, _mzd_addmul,
mzd_mul_m4rm, mzd_addmul_m4rm, _mzd_mul_m4rm, mzd_mul_mp, mzd_addmul_mp, squares with A == B), dimensions (log-uniform, pulled to
word / tile / Strassen-block boundaries and one off them half of the time; a quarter tiny, 7 % up to twice max_dim, 7 % on the leaf
shapes the rank-47 scheme passes take), operands as plain matrices or as windows of larger
parents (row / word-column offsets, excess bits, the neighbours' bits in the padding), a cutoff (0 or a power-of-two-ish hint), the
fused-pass depth (m4ri_amd_set_max_fuse 1..4) and the small-product rule (forced to the GPU, or the library's own).  After the call:
the result equals the reference's, the operands are untouched, and every bit of C's parent outside the window is what it was.
Prints one line per mismatch (with everything needed to replay it) and a summary; exit code 1 on any mismatch.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

import m4ri_amd  # noqa: E402
from m4ri_amd.mzd import Mzd  # noqa: E402
import cpu_libs  # noqa: E402

OPS = ["mul_null", "mul_dirty", "addmul", "mul_even", "addmul_even", "_addmul", "mul_m4rm", "addmul_m4rm", "_mul_m4rm0", "_mul_m4rm1",
       "mul_mp", "addmul_mp", "sqr", "addsqr"]


def draw_dim(rng, hi, lo=1):
    """log-uniform in [lo, hi], half of the time snapped to a boundary the kernels care about, +-1 around it a third of those."""
    x = int(np.exp(rng.uniform(np.log(lo), np.log(hi))))
    if rng.random() < 0.5:
        g = int(rng.choice([64, 128, 256, 512, 1024, 2048, 4096]))
        x = max(g, (x + g // 2) // g * g)
        if rng.random() < 0.33:
            x += int(rng.choice([-1, 1, 63, -63]))
    return max(1, min(x, hi))


def operand(rng, rows, cols, seed, window):
    """rows x cols matrix with splitmix bits; as a window it sits inside a parent with random bits all around it."""
    if not window:
        return Mzd.random(rows, cols, seed), None
    r0, c0 = int(rng.integers(0, 70)), 64 * int(rng.integers(0, 4))
    extra_r, extra_c = int(rng.integers(0, 70)), int(rng.integers(0, 200))
    P = Mzd.random(r0 + rows + extra_r, c0 + cols + extra_c, seed)
    return P.window(r0, c0, r0 + rows, c0 + cols), P


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else 9000
    rng = np.random.default_rng(seed)
    trace = bool(os.environ.get("SOAK_TRACE"))   # one line per case BEFORE it runs (what a crash needs)
    import faulthandler
    faulthandler.enable()
    ref = cpu_libs.reference()
    chk = ref if ref is not None else cpu_libs.oracle()
    kind = "reference (oracle/_ref)" if ref is not None else "oracle (oracle/_ref absent)"
    m4ri_amd.init(0)
    default_small = m4ri_amd.set_small_product_threshold(0)
    t0, cases, bad, by_op, bitops = time.time(), 0, 0, {}, 0.0
    while time.time() - t0 < budget:
        op = OPS[int(rng.integers(0, len(OPS)))]
        cls = rng.random()
        if cls < 0.25:    # tiny: word and tile edges, the small-product rule
            m, l, n = (draw_dim(rng, 300) for _ in range(3))
        elif cls < 0.86:  # the bulk
            m, l, n = (draw_dim(rng, hi, 48) for _ in range(3))
        elif cls < 0.93:  # large
            m, l, n = (draw_dim(rng, 2 * hi, 512) for _ in range(3))
        else:             # leaf shapes the rank-47 scheme passes take (scheme_passes.hip: leaf l % 1024 == 0, leaf n % 4096 == 0), +- strips
            m = 128 * int(rng.integers(2, 129))
            l = 4096 * int(rng.integers(1, 4))
            n = 16384
            if rng.random() < 0.4:
                m, l, n = m + int(rng.integers(0, 70)), l + int(rng.integers(0, 130)), n + int(rng.integers(0, 130))
        if cls < 0.93 and rng.random() < 0.1:  # thin in one dimension
            which = int(rng.integers(0, 3))
            m, l, n = [(int(rng.integers(1, 65)) if i == which else v) for i, v in enumerate((m, l, n))]
        if op in ("sqr", "addsqr"):
            l = n = m
        if float(m) * l * n > 4.0e12:  # keep one reference product below ~ 1.5 s
            continue
        cutoff = 0 if rng.random() < 0.5 else int(rng.choice([64, 128, 256, 512, 1024, 2048, 4096, 1000, 3000]))
        if cls >= 0.93:
            cutoff = int(rng.choice([0, 1024, 2048, 4096]))
        k = 0 if rng.random() < 0.7 else int(rng.integers(1, 17))
        fuse = int(rng.choice([0, 0, 1, 2, 3, 4]))
        small = rng.random() < 0.2
        wa, wb, wc = (rng.random() < 0.3 for _ in range(3))
        sa, sb, sc = (int(x) for x in rng.integers(1, 1 << 40, size=3))
        A, PA = operand(rng, m, l, sa, wa)
        if op in ("sqr", "addsqr"):
            B, PB = A, PA
        else:
            B, PB = operand(rng, l, n, sb, wb)
        needs_c = op != "mul_null"
        C, PC = operand(rng, m, n, sc, wc) if needs_c else (None, None)
        if trace:
            print(f"case op={op} m={m} l={l} n={n} cutoff={cutoff} k={k} fuse={fuse} small={small} windows={wa, wb, wc} seeds={sa, sb, sc}", flush=True)
        a0, b0 = A.masked().copy(), B.masked().copy()
        pc0 = PC.rows().copy() if PC is not None else None
        add = op in ("addmul", "addmul_even", "_addmul", "addmul_m4rm", "_mul_m4rm0", "addmul_mp", "addsqr")
        # the reference on copies (plain matrices: the reference's own window handling is not what is under test)
        Ar, Br = A.copy(), B.copy() if B is not A else None
        if Br is None:
            Br = Ar
        if op in ("mul_m4rm", "addmul_m4rm", "_mul_m4rm0", "_mul_m4rm1"):
            want = chk.addmul_m4rm(C.copy(), Ar, Br, k) if add and hasattr(chk, "addmul_m4rm") else \
                (chk.addmul(C.copy(), Ar, Br, 0) if add else chk.mul_m4rm(None, Ar, Br, k))
        else:
            # The checker is always called with cutoff 0 (its default, 4096): a cutoff is a hint that changes no bit (pinned by the oracle
            # tests), and the reference's own recursion is not safe for every hint -- with cutoff 64 a dimension in 86 .. 127 gives it empty
            # quadrants, which _mzd_addmul_even (strassen.c:396-420) and _mzd_sqr_even (:210-240) pass on to the M4RM leaf: it aborts in
            # mzd_copy ("Target matrix is too small") or reads out of bounds.  The library under test gets the drawn cutoff.
            want = chk.addmul(C.copy(), Ar, Br, 0) if add else chk.mul(None, Ar, Br, 0)
        m4ri_amd.set_max_fuse(fuse)
        m4ri_amd.set_small_product_threshold(default_small if small else 0)
        ec = max(64, cutoff)
        L = m4ri_amd.lib()
        if op == "mul_null":
            got = m4ri_amd.mzd_mul(None, A, B, cutoff)
        elif op == "mul_dirty":
            got = m4ri_amd.mzd_mul(C, A, B, cutoff)
        elif op == "addmul":
            got = m4ri_amd.mzd_addmul(C, A, B, cutoff)
        elif op == "mul_even":
            got = m4ri_amd._mzd_mul_even(C, A, B, ec)
        elif op == "addmul_even":
            got = m4ri_amd._mzd_addmul_even(C, A, B, ec)
        elif op == "_addmul":
            got = m4ri_amd._mzd_addmul(C, A, B, ec)
        elif op == "mul_m4rm":
            got = m4ri_amd.mzd_mul_m4rm(C, A, B, k)
        elif op == "addmul_m4rm":
            got = m4ri_amd.mzd_addmul_m4rm(C, A, B, k)
        elif op == "_mul_m4rm0":
            got = m4ri_amd._mzd_mul_m4rm(C, A, B, k, 0)
        elif op == "_mul_m4rm1":
            got = m4ri_amd._mzd_mul_m4rm(C, A, B, k, 1)
        elif op == "mul_mp":
            got = m4ri_amd.mzd_mul_mp(C, A, B, cutoff)
        elif op == "addmul_mp":
            got = m4ri_amd.mzd_addmul_mp(C, A, B, cutoff)
        elif op == "sqr":
            L._mzd_sqr_even(C.ptr, A.ptr, ec)
            got = C
        else:
            L._mzd_addsqr_even(C.ptr, A.ptr, ec)
            got = C
        ok = got.equal(want) and np.array_equal(A.masked(), a0) and np.array_equal(B.masked(), b0)
        if ok and PC is not None:  # every bit of the parent outside the window's valid columns is untouched
            now = PC.rows().copy()
            r0 = (C.offset - PC.offset) // PC.rowstride
            c0 = (C.offset - PC.offset) % PC.rowstride
            inside = np.zeros_like(now, dtype=bool)
            inside[r0:r0 + m, c0:c0 + C.width] = True
            ok = np.array_equal(now[~inside], pc0[~inside])
            if ok and n % 64:  # the excess bits of the window's last word belong to the parent
                mask = ~np.uint64(C.high_bitmask)
                ok = np.array_equal(now[r0:r0 + m, c0 + C.width - 1] & mask, pc0[r0:r0 + m, c0 + C.width - 1] & mask)
        cases += 1
        bitops += float(m) * l * n
        by_op[op] = by_op.get(op, 0) + 1
        if not ok:
            bad += 1
            print(f"MISMATCH op={op} m={m} l={l} n={n} cutoff={cutoff} k={k} fuse={fuse} small={small} windows={wa, wb, wc} seeds={sa, sb, sc} soak_seed={seed} case={cases}",
                  flush=True)
    m4ri_amd.set_max_fuse(0)
    m4ri_amd.set_small_product_threshold(default_small)
    print(f"soak_mul seed {seed}: {cases} cases in {time.time() - t0:.0f} s against the {kind}, {bad} mismatches, {bitops:.3g} bit-ops checked, "
          f"dims <= {hi} (7 % up to {2 * hi}, 7 % scheme shapes up to 16384 x 12288 x 16384); by entry point: " + ", ".join(f"{k} {v}" for k, v in sorted(by_op.items())), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
