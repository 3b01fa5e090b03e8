#!/usr/bin/env python3
"""Differential soak of the multi-device path on virtual ranks (a script, not a pytest module; it lives under tests/ because it runs the
checker; GPU box): random world sizes, shapes, layouts, schedules and sharded depths through libm4ri_amd.so's part-4 entry points
against the REAL reference built into oracle/_ref (the CPU oracle when that build is absent), bit for bit, for a wall-clock budget.

    python tests/soak_multi.py [seconds] [seed] [max_dim]

Every case draws a world size (2 .. 8 ranks sharing device 0: own streams, slabs, link streams and host threads, pieces by peer copies
device 0 -> device 0), dimensions (log-uniform, pulled to word / tile / block boundaries half of the time; 7 % of the cases with l and n
at 16384, where two sharded levels are ONE application of the rank-47 scheme: 47 sub-products, grouped into batched products), and one of:
  mul_multi    m4ri_amd_mul_multi on host matrices (plain or windows), forced sharded depth 0 / 1 / 2, mul or addmul
  mp           mzd_mul_mp / mzd_addmul_mp with the size threshold lowered so that the product is spread
  dmat         m4ri_amd_dmat_mul on distributed resident matrices: random layouts of A, B and C (rows, 1- and 2-level slab-cyclic,
               replicated), schedule auto / slabs / strassen, mul or addmul, and a second product chained on the result
After the call the result equals the reference's and every bit of C's parent outside a window is what it was.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401 -- before libm4ri_amd.so: the process gets ONE HIP runtime, the one torch ships

import m4ri_amd  # noqa: E402
from m4ri_amd import Dmat  # noqa: E402
from m4ri_amd.mzd import Mzd  # noqa: E402
import cpu_libs  # noqa: E402
from soak_mul import draw_dim, operand  # noqa: E402

LAYOUTS = [m4ri_amd.LAYOUT_ROWS, m4ri_amd.LAYOUT_CYCLIC1, m4ri_amd.LAYOUT_CYCLIC2, m4ri_amd.LAYOUT_REPLICATED]
VARIANTS = [m4ri_amd.VARIANT_AUTO, m4ri_amd.VARIANT_SLABS, m4ri_amd.VARIANT_STRASSEN]


def parent_untouched(PC, C, pc0, m, n):
    now = PC.rows().copy()
    r0, c0 = (C.offset - PC.offset) // PC.rowstride, (C.offset - PC.offset) % PC.rowstride
    inside = np.zeros_like(now, dtype=bool)
    inside[r0:r0 + m, c0:c0 + C.width] = True
    ok = np.array_equal(now[~inside], pc0[~inside])
    if ok and n % 64:
        mask = ~np.uint64(C.high_bitmask)
        ok = np.array_equal(now[r0:r0 + m, c0 + C.width - 1] & mask, pc0[r0:r0 + m, c0 + C.width - 1] & mask)
    return ok


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else 6000
    rng = np.random.default_rng(seed)
    trace = bool(os.environ.get("SOAK_TRACE"))
    import faulthandler
    faulthandler.enable()
    ref = cpu_libs.reference()
    chk = ref if ref is not None else cpu_libs.oracle()
    kind = "reference (oracle/_ref)" if ref is not None else "oracle (oracle/_ref absent)"
    m4ri_amd.init(0)
    m4ri_amd.set_small_product_threshold(0)
    t0, cases, bad, by, bitops = time.time(), 0, 0, {}, 0.0
    while time.time() - t0 < budget:
        world = int(rng.choice([2, 3, 4, 5, 6, 7, 8]))
        what = str(rng.choice(["mul_multi", "mp", "dmat", "dmat"]))
        cls = rng.random()
        lo = 1 if cls < 0.2 else 48
        lim = 300 if cls < 0.2 else (2 * hi if cls > 0.93 else hi)
        m, l, n = (draw_dim(rng, lim, lo) for _ in range(3))
        split47 = cls > 0.86 and cls <= 0.93   # two sharded levels as ONE application of the rank-47 scheme: padded l and n multiples of 16384
        if split47:
            m = int(rng.integers(4 * world, 8193)) if rng.random() < 0.5 else 2048 * int(rng.integers(1, 5))
            l, n = (16384 - (int(rng.integers(0, 256)) if rng.random() < 0.4 else 0) for _ in range(2))
            what = str(rng.choice(["mul_multi", "dmat"]))
        if float(m) * l * n > 2.3e12:
            continue
        add = bool(rng.random() < 0.5)
        cutoff = 0 if rng.random() < 0.6 else int(rng.choice([64, 256, 1024, 2048]))
        sa, sb, sc = (int(x) for x in rng.integers(1, 1 << 40, size=3))
        tag = what
        m4ri_amd.set_devices([0] * world)
        if what == "dmat":
            la, lb, lc = (int(rng.choice(LAYOUTS)) for _ in range(3))
            if lc == m4ri_amd.LAYOUT_REPLICATED:
                lc = m4ri_amd.LAYOUT_ROWS
            variant = int(rng.choice(VARIANTS))
            if split47:
                variant = m4ri_amd.VARIANT_STRASSEN
                if rng.random() < 0.6:
                    la = lb = lc = m4ri_amd.LAYOUT_CYCLIC2
            tag = f"dmat la={la} lb={lb} lc={lc} variant={variant}"
        elif what == "mul_multi":
            levels = 2 if split47 else int(rng.integers(0, 3))
            wa, wb, wc = (bool(rng.random() < 0.3) for _ in range(3))
            tag = f"mul_multi levels={levels} windows={wa, wb, wc}"
        if trace:
            print(f"case world={world} {tag} m={m} l={l} n={n} add={add} cutoff={cutoff} seeds={sa, sb, sc}", flush=True)
        ok = True
        if what == "dmat":
            A, B, C0 = Mzd.random(m, l, sa), Mzd.random(l, n, sb), Mzd.random(m, n, sc)
            want = chk.addmul(C0.copy(), A, B, 0) if add else chk.mul(None, A, B, 0)
            dA, dB, dC = Dmat(m, l, la).upload(A), Dmat(l, n, lb).upload(B), Dmat(m, n, lc).upload(C0)
            m4ri_amd.dmat_mul(dC, dA, dB, add, cutoff, variant)
            ok = dC.download().equal(want) and dA.download().equal(A) and dB.download().equal(B)
            st = m4ri_amd.multi_stats()
            if st.variant == m4ri_amd.VARIANT_STRASSEN and st.levels == 2:
                by[f"dmat_sub_products_{st.sub_products}"] = by.get(f"dmat_sub_products_{st.sub_products}", 0) + 1
            if ok and rng.random() < 0.4:  # the result as the left operand of a second product, in the layout it was left in
                k2 = draw_dim(rng, min(hi, 2000), 1)
                E = Mzd.random(n, k2, sa ^ 0x55)
                dE, dF = Dmat(n, k2, int(rng.choice(LAYOUTS))).upload(E), Dmat(m, k2, lc)
                m4ri_amd.dmat_mul(dF, dC, dE, False, 0, m4ri_amd.VARIANT_AUTO)
                ok = dF.download().equal(chk.mul(None, want, E, 0))
                dE.free()
                dF.free()
            for d in (dA, dB, dC):
                d.free()
        elif what == "mp":
            old = m4ri_amd.set_multi_threshold(int(rng.choice([1, 64, 512])))
            A, B, C0 = Mzd.random(m, l, sa), Mzd.random(l, n, sb), Mzd.random(m, n, sc)
            want = chk.addmul(C0.copy(), A, B, 0) if add else chk.mul(None, A, B, 0)
            got = m4ri_amd.mzd_addmul_mp(C0, A, B, cutoff) if add else m4ri_amd.mzd_mul_mp(C0 if rng.random() < 0.5 else None, A, B, cutoff)
            ok = got.equal(want)
            m4ri_amd.set_multi_threshold(old)
        else:
            (A, _), (B, _) = operand(rng, m, l, sa, wa), operand(rng, l, n, sb, wb)
            C, PC = operand(rng, m, n, sc, wc)
            pc0 = PC.rows().copy() if PC is not None else None
            want = chk.addmul(C.copy(), A.copy(), B.copy(), 0) if add else chk.mul(None, A.copy(), B.copy(), 0)
            a0, b0 = A.masked().copy(), B.masked().copy()
            got = m4ri_amd.mul_multi(C, A, B, add, cutoff, levels)
            st = m4ri_amd.multi_stats()
            if st.variant == m4ri_amd.VARIANT_STRASSEN and st.levels == 2:
                by[f"mul_multi_sub_products_{st.sub_products}"] = by.get(f"mul_multi_sub_products_{st.sub_products}", 0) + 1
            ok = got.equal(want) and np.array_equal(A.masked(), a0) and np.array_equal(B.masked(), b0)
            if ok and PC is not None:
                ok = parent_untouched(PC, C, pc0, m, n)
        cases += 1
        bitops += float(m) * l * n
        by[what] = by.get(what, 0) + 1
        if not ok:
            bad += 1
            print(f"MISMATCH world={world} {tag} m={m} l={l} n={n} add={add} cutoff={cutoff} seeds={sa, sb, sc} soak_seed={seed} case={cases}", flush=True)
    m4ri_amd.set_devices([])
    print(f"soak_multi seed {seed}: {cases} cases in {time.time() - t0:.0f} s against the {kind}, {bad} mismatches, {bitops:.3g} bit-ops checked, worlds 2 .. 8 "
          f"(virtual ranks on device 0), dims <= {hi} (7 % up to {2 * hi}); " + ", ".join(f"{k} {v}" for k, v in sorted(by.items())), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
