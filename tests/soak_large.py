#!/usr/bin/env python3
"""Differential soak at LARGE ragged sizes (a script, not a pytest module; it lives under tests/ because it runs the checker; GPU box):
random products with dimensions up to 70000 -- where row blocks, remainder strips, depth choices, 32-bit offsets and the host
block pipeline all come into play -- through libm4ri_amd.so's M4RI-named entry points against the REAL reference's multi-core path
(oracle/_ref's OpenMP build, mzd_mul_mp / mzd_addmul_mp on every host core), bit for bit, for a wall-clock budget.

    python tests/soak_large.py [seconds] [seed] [max_dim]

Every case draws dimensions (log-uniform in 6000 .. max_dim, pulled to a 4096-boundary +- a few bits a third of the time, at most
65536^3 bit-ops), mul or addmul, operands as plain matrices or windows of wider / taller parents (row strides that are not the width),
and how the operands reach the device: pageable host matrices (the block pipeline of mzd_api.hip) or pinned ones (m4ri_amd_pin: the
resident path, result synced back).  After the call: the result equals the reference's, operands untouched, C's parent intact.
"""
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", str(os.cpu_count() or 1))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

import m4ri_amd  # noqa: E402
from m4ri_amd.mzd import Mzd  # noqa: E402
import cpu_libs  # noqa: E402
from soak_multi import parent_untouched  # noqa: E402


def draw(rng, lo, hi):
    x = int(np.exp(rng.uniform(np.log(lo), np.log(hi))))
    if rng.random() < 0.33:
        x = max(4096, (x + 2048) // 4096 * 4096) + int(rng.choice([0, 0, -1, 1, 63, -65, 130]))
    return max(lo, min(x, hi))


def operand(rng, rows, cols, seed, window):
    if not window:
        return Mzd.random(rows, cols, seed), None
    r0, c0 = int(rng.integers(0, 40)), 64 * int(rng.integers(0, 3))
    P = Mzd.random(r0 + rows + int(rng.integers(0, 40)), c0 + cols + int(rng.integers(0, 300)), seed)
    return P.window(r0, c0, r0 + rows, c0 + cols), P


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else 70000
    rng = np.random.default_rng(seed)
    import faulthandler
    faulthandler.enable()
    omp = cpu_libs.reference(openmp=True)
    seq = cpu_libs.reference()
    assert omp is not None or seq is not None, "oracle/_ref is not built: this soak needs the real reference"
    m4ri_amd.init(0)
    t0, cases, bad, bitops, t_ref, t_gpu = time.time(), 0, 0, 0.0, 0.0, 0.0
    while time.time() - t0 < budget:
        m, l, n = draw(rng, 6000, hi), draw(rng, 6000, hi), draw(rng, 6000, hi)
        if rng.random() < 0.2:  # one thin dimension against two large ones
            which = int(rng.integers(0, 3))
            m, l, n = [(int(rng.integers(64, 3000)) if i == which else v) for i, v in enumerate((m, l, n))]
        if float(m) * l * n > 2.9e14:
            continue
        add = bool(rng.random() < 0.5)
        pinned = bool(rng.random() < 0.5)
        wa, wb, wc = (bool(rng.random() < 0.3) for _ in range(3))
        sa, sb, sc = (int(x) for x in rng.integers(1, 1 << 40, size=3))
        print(f"case m={m} l={l} n={n} add={add} pinned={pinned} windows={wa, wb, wc} seeds={sa, sb, sc}", flush=True)
        (A, PA), (B, PB), (C, PC) = operand(rng, m, l, sa, wa), operand(rng, l, n, sb, wb), operand(rng, m, n, sc, wc)
        # (C = A*B on the reference's multi-core path needs a ZERO result block: _mzd_mul_mp4, mp.c:212-235, adds the remainder strips
        # of a ragged product onto whatever C held -- a defect of its own, documented C = AB in mp.h:34-47; the library overwrites)
        Ar, Br, Cr = A.copy(), B.copy(), (C.copy() if add else Mzd.init(m, n))
        t = time.time()
        if omp is not None and omp.has_mp:
            want = omp.L.mzd_addmul_mp(Cr.ptr, Ar.ptr, Br.ptr, 0) if add else omp.L.mzd_mul_mp(Cr.ptr, Ar.ptr, Br.ptr, 0)
            want = Cr
        else:
            want = seq.addmul(Cr, Ar, Br, 0) if add else seq.mul(Cr, Ar, Br, 0)
        t_ref += time.time() - t
        a0, b0 = A.masked().copy(), B.masked().copy()
        pc0 = PC.rows().copy() if PC is not None else None
        t = time.time()
        pins = [P if P is not None else M for (M, P) in ((A, PA), (B, PB), (C, PC))] if pinned else []
        for M in pins:
            m4ri_amd.pin(M)
        got = m4ri_amd.mzd_addmul(C, A, B, 0) if add else m4ri_amd.mzd_mul(C, A, B, 0)
        for M in pins:
            m4ri_amd.unpin(M)   # (a pinned result is synced back by unpin)
        t_gpu += time.time() - t
        ok = got.equal(want) and np.array_equal(A.masked(), a0) and np.array_equal(B.masked(), b0)
        if ok and PC is not None:
            ok = parent_untouched(PC, C, pc0, m, n)
        cases += 1
        bitops += float(m) * l * n
        if not ok:
            bad += 1
            print(f"MISMATCH m={m} l={l} n={n} add={add} pinned={pinned} windows={wa, wb, wc} seeds={sa, sb, sc} soak_seed={seed}", flush=True)
    print(f"soak_large seed {seed}: {cases} cases in {time.time() - t0:.0f} s against the reference's multi-core path ({os.environ['OMP_NUM_THREADS']} threads), "
          f"{bad} mismatches, {bitops:.3g} bit-ops checked, dims 6000 .. {hi}; reference {t_ref:.0f} s, library incl. pin / transfers {t_gpu:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
