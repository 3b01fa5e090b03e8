#!/usr/bin/env python3
"""Differential soak of the DEVICE-pointer entry points (a script, not a pytest module; it lives under tests/ because it runs the
checker; GPU box): m4ri_amd_mul_dev, m4ri_amd_mul_batch_dev, m4ri_amd_m4rm_dev and m4ri_amd_m4rm_batch_dev on operands resident in HBM --
random shapes, row strides wider than the width, batch strides with gaps, batch sizes, accumulate or not, cutoffs that force one to four
Strassen-Winograd levels at small sizes, the fused-pass depth, the leaf's inner-dimension split -- against the REAL reference built into
oracle/_ref (the CPU oracle when that build is absent), bit for bit, gaps and padding words included, for a wall-clock budget.

    python tests/soak_dev.py [seconds] [seed] [max_dim]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import m4ri_amd  # noqa: E402
from m4ri_amd.mzd import Mzd  # noqa: E402
import cpu_libs  # noqa: E402
from soak_mul import draw_dim  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
    rng = np.random.default_rng(seed)
    trace = bool(os.environ.get("SOAK_TRACE"))
    import faulthandler
    faulthandler.enable()
    ref = cpu_libs.reference()
    chk = ref if ref is not None else cpu_libs.oracle()
    kind = "reference (oracle/_ref)" if ref is not None else "oracle (oracle/_ref absent)"
    m4ri_amd.init(0)
    torch.cuda.set_device(0)
    L = m4ri_amd.lib()
    t0, cases, bad, by, bitops = time.time(), 0, 0, {}, 0.0
    while time.time() - t0 < budget:
        what = str(rng.choice(["mul_dev", "mul_batch_dev", "mul_batch_dev", "m4rm_dev", "m4rm_batch_dev"]))
        cls = rng.random()
        m, l, n = (draw_dim(rng, 300 if cls < 0.2 else hi, 1 if cls < 0.2 else 48) for _ in range(3))
        if cls > 0.8:  # even blocks: no strips, so a batch really runs as one product (engine_mul_batch)
            g = int(rng.choice([256, 512, 1024]))
            m, l, n = (max(g, x // g * g) for x in (m, l, n))
        batch = 1 if what in ("mul_dev", "m4rm_dev") else int(rng.integers(1, 9))
        if float(m) * l * n * batch > 1.5e12:
            continue
        add = bool(rng.random() < 0.5)
        cutoff = 0 if rng.random() < 0.35 else int(rng.choice([64, 128, 256, 512, 1024]))
        fuse = int(rng.choice([0, 0, 1, 2, 3, 4]))
        ksplit = int(rng.choice([0, 0, 1, 2, 3, 8]))
        wl, wn = (l + 63) // 64, (n + 63) // 64
        sa, sb, sc = wl + int(rng.integers(0, 4)), wn + int(rng.integers(0, 4)), wn + int(rng.integers(0, 6))
        abs_, bbs, cbs = m * sa + int(rng.integers(0, 40)), l * sb + int(rng.integers(0, 40)), m * sc + int(rng.integers(0, 40))
        seeds = [int(x) for x in rng.integers(1, 1 << 40, size=3)]
        if trace:
            print(f"case {what} m={m} l={l} n={n} batch={batch} add={add} cutoff={cutoff} fuse={fuse} ksplit={ksplit} strides={sa, sb, sc} seeds={seeds}", flush=True)
        A = [Mzd.random(m, l, seeds[0] + b) for b in range(batch)]
        B = [Mzd.random(l, n, seeds[1] + b) for b in range(batch)]
        C = [Mzd.random(m, n, seeds[2] + b) for b in range(batch)]
        # every word of the buffers is random: gaps, stride padding and the excess bits of the last word (the operands' must not matter, C's
        # gaps must survive; the library defines C's excess bits of the last valid word as zero after a device product -- compared masked)
        hA = rng.integers(0, 1 << 63, size=batch * abs_ + 8, dtype=np.int64).view(np.uint64)
        hB = rng.integers(0, 1 << 63, size=batch * bbs + 8, dtype=np.int64).view(np.uint64)
        hC = rng.integers(0, 1 << 63, size=batch * cbs + 8, dtype=np.int64).view(np.uint64)
        for b in range(batch):
            va = hA[b * abs_: b * abs_ + m * sa].reshape(m, sa)
            va[:, :wl] = A[b].masked()            # device operands keep the bits beyond ncols zero
            vb = hB[b * bbs: b * bbs + l * sb].reshape(l, sb)
            vb[:, :wn] = B[b].masked()
            vc = hC[b * cbs: b * cbs + m * sc].reshape(m, sc)
            vc[:, :wn] = C[b].masked()
        keep = hC.copy()
        tA, tB, tC = (torch.from_numpy(x.view(np.int64)).cuda() for x in (hA, hB, hC))
        m4ri_amd.set_max_fuse(fuse)
        if what == "mul_dev":
            m4ri_amd.mul_dev(tC.data_ptr(), sc, tA.data_ptr(), sa, tB.data_ptr(), sb, m, l, n, add, cutoff, 0)
        elif what == "mul_batch_dev":
            m4ri_amd.mul_batch_dev(tC.data_ptr(), sc, cbs, tA.data_ptr(), sa, abs_, tB.data_ptr(), sb, bbs, m, l, n, batch, add, cutoff, 0)
        elif what == "m4rm_dev":
            m4ri_amd.m4rm_dev(tC.data_ptr(), sc, tA.data_ptr(), sa, tB.data_ptr(), sb, m, l, n, add, ksplit, 0)
        else:
            rc = L.m4ri_amd_m4rm_batch_dev(tC.data_ptr(), sc, cbs, tA.data_ptr(), sa, abs_, tB.data_ptr(), sb, bbs, m, l, n, batch, int(add), None)
            assert rc == 0, rc
        torch.cuda.synchronize()
        got = tC.cpu().numpy().view(np.uint64)
        ok = np.array_equal(tA.cpu().numpy().view(np.uint64), hA) and np.array_equal(tB.cpu().numpy().view(np.uint64), hB)
        outside = np.ones(got.shape, dtype=bool)
        for b in range(batch):
            want = chk.addmul(C[b].copy(), A[b], B[b], 0) if add else chk.mul(None, A[b], B[b], 0)
            ok = ok and np.array_equal(got[b * cbs: b * cbs + m * sc].reshape(m, sc)[:, :wn], want.masked())
            outside[b * cbs: b * cbs + m * sc].reshape(m, sc)[:, :wn] = False
        ok = ok and np.array_equal(got[outside], keep[outside])   # stride padding, gaps between the batch's members, the tail
        cases += 1
        bitops += float(m) * l * n * batch
        by[what] = by.get(what, 0) + 1
        if not ok:
            bad += 1
            print(f"MISMATCH {what} m={m} l={l} n={n} batch={batch} add={add} cutoff={cutoff} fuse={fuse} ksplit={ksplit} strides={sa, sb, sc} "
                  f"bs={abs_, bbs, cbs} seeds={seeds} soak_seed={seed} case={cases}", flush=True)
    m4ri_amd.set_max_fuse(0)
    print(f"soak_dev seed {seed}: {cases} cases in {time.time() - t0:.0f} s against the {kind}, {bad} mismatches, {bitops:.3g} bit-ops checked, dims <= {hi}, "
          f"batches of 1 .. 8; " + ", ".join(f"{k} {v}" for k, v in sorted(by.items())), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
