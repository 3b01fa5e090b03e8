"""CPU stand-ins for the three device steps of the Strassen-sharded product (tests only): the Winograd
down / up passes of aux_kernels.hip restated in numpy on local parents, and the sub-products done by the
oracle.  The plan, the piece table and the exchange logic under test are the product's own
(m4ri_amd_shard_* in libm4ri_amd.so, m4ri_amd/sharding.py)."""
import numpy as np

import m4ri_amd
from m4ri_amd import sharding
from m4ri_amd.mzd import Mzd


def down(X, bside, levels):
    """X: (2^levels * s) x (2^levels * cw) words -> list of 7^levels children (s x cw), index 7*j1 + j2."""
    if levels == 0:
        return [X]
    r, c = X.shape[0] // 2, X.shape[1] // 2
    x11, x12, x21, x22 = X[:r, :c], X[:r, c:], X[r:, :c], X[r:, c:]
    if not bside:  # aux_kernels.hip: [A11, A12, S4, A22, S1, S2, S3]
        s1 = x21 ^ x22; s2 = s1 ^ x11; s3 = x11 ^ x21; s4 = x12 ^ s2
        ch = [x11, x12, s4, x22, s1, s2, s3]
    else:          # [B11, B21, B22, T4, T1, T2, T3]
        t1 = x12 ^ x11; t2 = x22 ^ t1; t3 = x22 ^ x12; t4 = t2 ^ x21
        ch = [x11, x21, x22, t4, t1, t2, t3]
    out = []
    for c_ in ch:
        out.extend(down(c_, bside, levels - 1))
    return out


def up(P, levels):
    """7^levels products (s x cw each, index 7*j1 + j2) -> the (2^levels * s) x (2^levels * cw) parent."""
    if levels == 0:
        return P[0]
    k = len(P) // 7
    p = [up(P[i * k:(i + 1) * k], levels - 1) for i in range(7)]
    u2 = p[0] ^ p[5]; u3 = u2 ^ p[6]; u4 = u2 ^ p[4]
    return np.block([[p[0] ^ p[1], u4 ^ p[2]], [u3 ^ p[3], u3 ^ p[4]]])


_SCHEME = None


def scheme_tables():
    """(U, V, W) of m4ri_amd/csrc/scheme444.h: product r = (sum of blocks A_ij with bit 4 i + j of U[r]) * (sum of B_jk with bit 4 j + k of
    V[r]); C_ik = sum of the products r with bit 4 i + k of W[r] (the table test_host_logic.py verifies against the definition)."""
    global _SCHEME
    if _SCHEME is None:
        import os
        import re
        text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "m4ri_amd", "csrc", "scheme444.h")).read()
        _SCHEME = tuple([int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", re.search(r"SCHEME444_%s\[SCHEME444_R\] = \{([^}]*)\}" % f, text).group(1))] for f in "UVW")
    return _SCHEME


def scheme_down(X, bside):
    """X: (4 s) x (4 cw) words -> the R operand sums of the 4 x 4 x 4 scheme (s x cw each), in the table's order (scheme_passes.hip)."""
    U, V, _ = scheme_tables()
    r, c = X.shape[0] // 4, X.shape[1] // 4
    out = []
    for mask in (V if bside else U):
        acc = np.zeros((r, c), dtype=np.uint64)
        for f in range(16):
            if (mask >> f) & 1:
                acc ^= X[(f >> 2) * r:(f >> 2) * r + r, (f & 3) * c:(f & 3) * c + c]
        out.append(acc)
    return out


def scheme_up(P):
    """R products (s x cw each) -> the (4 s) x (4 cw) parent."""
    _, _, W = scheme_tables()
    r, c = P[0].shape
    out = np.zeros((4 * r, 4 * c), dtype=np.uint64)
    for mask, prod in zip(W, P):
        for f in range(16):
            if (mask >> f) & 1:
                out[(f >> 2) * r:(f >> 2) * r + r, (f & 3) * c:(f & 3) * c + c] ^= prod
    return out


def plan_down(plan, X, bside):
    """The plan's own operand pass: Winograd levels (7, 49 sub-products) or -- two levels, nprod != 49 -- the rank-R scheme once."""
    return scheme_down(X, bside) if (plan.levels == 2 and plan.nprod != 49) else down(X, bside, plan.levels)


def plan_up(plan, P):
    return scheme_up(P) if (plan.levels == 2 and plan.nprod != 49) else up(P, plan.levels)


def local_parent(plan, rank, which, M: Mzd, width_words):
    """The zero-padded local parent of host matrix M on `rank` (which: 0 rows over bm, 1 rows over bl)."""
    runs = sharding.local_rows(plan, rank, which)
    s = runs[0][1]
    out = np.zeros((plan.blocks * s, width_words), dtype=np.uint64)
    src = M.masked()
    for b, (g0, rows) in enumerate(runs):
        rows = max(0, min(rows, M.nrows - g0))
        if rows:
            out[b * s:b * s + rows, :M.width] = src[g0:g0 + rows]
    return out


def rank_part(plan, rank, A: Mzd, B: Mzd, oracle, exchange, chunks=1, group=1):
    """Rank `rank`'s whole part on CPU arrays; returns its local parent of C and its row runs."""
    LA = local_parent(plan, rank, 0, A, plan.L // 64)
    LB = local_parent(plan, rank, 1, B, plan.N // 64)
    names = {"child_a": m4ri_amd.BUF_CHILD_A, "child_b": m4ri_amd.BUF_CHILD_B, "slabs_p": m4ri_amd.BUF_SLABS_P,
             "oper_a": m4ri_amd.BUF_OPER_A, "oper_b": m4ri_amd.BUF_OPER_B, "prod": m4ri_amd.BUF_PROD}
    bufs = {k: np.zeros(max(1, m4ri_amd.shard_buffer_words(plan, rank, w)), dtype=np.uint64) for k, w in names.items()}
    sa, sb = m4ri_amd.shard_slab_rows(plan, rank, 0), m4ri_amd.shard_slab_rows(plan, rank, 1)
    out = {}

    def do_down():
        for key, X, bside, s, cw in (("child_a", LA, False, sa, plan.cwl), ("child_b", LB, True, sb, plan.cwn)):
            if s:
                ch = plan_down(plan, X, bside)
                bufs[key][:plan.nprod * s * cw] = np.concatenate([c.reshape(-1) for c in ch])

    def do_product(jl, j, row0=0, rows=None, w0=0, w1=None):
        rows = plan.bm if rows is None else rows
        w1 = plan.cwn if w1 is None else w1
        a0 = jl * plan.bm * plan.cwl + row0 * plan.cwl
        a = Mzd(rows, plan.cwl * 64, buf=bufs["oper_a"][a0:a0 + rows * plan.cwl], rowstride=plan.cwl)
        b = Mzd(plan.bl, (w1 - w0) * 64, buf=bufs["oper_b"], rowstride=plan.cwn, offset=jl * plan.bl * plan.cwn + w0, windowed=True)
        p = oracle.mul(None, a.copy(), b.copy(), 0)
        p0 = jl * plan.bm * plan.cwn + row0 * plan.cwn
        bufs["prod"][p0:p0 + rows * plan.cwn].reshape(rows, plan.cwn)[:, w0:w1] = p.masked()

    def do_up():
        if sa:
            P = [bufs["slabs_p"][j * sa * plan.cwn:(j + 1) * sa * plan.cwn].reshape(sa, plan.cwn) for j in range(plan.nprod)]
            out["C"] = plan_up(plan, P)
        else:
            out["C"] = np.zeros((0, plan.N // 64), dtype=np.uint64)

    def copy_local(dst, src):
        dst[...] = src

    def do_product_group(q0, count):   # what m4ri_amd_mul_batch_dev does on the device: the rank's sub-products q0 .. q0 + count - 1
        owned = sharding.owned_products(plan, rank)
        for q in range(q0, q0 + count):
            do_product(q, owned[q])

    sharding.run_strassen_sharded(plan, rank, bufs, do_down, do_product, do_up, exchange, copy_local, chunks=chunks, group=group,
                                  product_group=do_product_group if group > 1 else None)
    return out["C"], sharding.local_rows(plan, rank, 0)


def rank_products(plan, rank, pairs, oracle, exchange, inflight=2, chunks=1):
    """Several products [(A_k, B_k)] through sharding.run_products on `inflight` buffer slots (product k on slot k % inflight):
    rank `rank`'s local parents of every C_k.  What is under test is that a slot's buffers are not reused before their product
    is through with them."""
    names = {"child_a": m4ri_amd.BUF_CHILD_A, "child_b": m4ri_amd.BUF_CHILD_B, "slabs_p": m4ri_amd.BUF_SLABS_P,
             "oper_a": m4ri_amd.BUF_OPER_A, "oper_b": m4ri_amd.BUF_OPER_B, "prod": m4ri_amd.BUF_PROD}
    slots = [{k: np.zeros(max(1, m4ri_amd.shard_buffer_words(plan, rank, w)), dtype=np.uint64) for k, w in names.items()} for _ in range(max(1, inflight))]
    sa, sb = m4ri_amd.shard_slab_rows(plan, rank, 0), m4ri_amd.shard_slab_rows(plan, rank, 1)
    out = {}

    def make_step(k):
        bufs = slots[k % len(slots)]
        A, B = pairs[k]
        LA = local_parent(plan, rank, 0, A, plan.L // 64)
        LB = local_parent(plan, rank, 1, B, plan.N // 64)

        def do_down():
            for key, X, bside, s_, cw in (("child_a", LA, False, sa, plan.cwl), ("child_b", LB, True, sb, plan.cwn)):
                if s_:
                    bufs[key][:plan.nprod * s_ * cw] = np.concatenate([c.reshape(-1) for c in plan_down(plan, X, bside)])

        def do_product(jl, j, row0=0, rows=None, w0=0, w1=None):
            rows = plan.bm if rows is None else rows
            w1 = plan.cwn if w1 is None else w1
            a0 = jl * plan.bm * plan.cwl + row0 * plan.cwl
            a = Mzd(rows, plan.cwl * 64, buf=bufs["oper_a"][a0:a0 + rows * plan.cwl], rowstride=plan.cwl)
            b = Mzd(plan.bl, (w1 - w0) * 64, buf=bufs["oper_b"], rowstride=plan.cwn, offset=jl * plan.bl * plan.cwn + w0, windowed=True)
            p0 = jl * plan.bm * plan.cwn + row0 * plan.cwn
            bufs["prod"][p0:p0 + rows * plan.cwn].reshape(rows, plan.cwn)[:, w0:w1] = oracle.mul(None, a.copy(), b.copy(), 0).masked()

        def do_up():
            if sa:
                P = [bufs["slabs_p"][j * sa * plan.cwn:(j + 1) * sa * plan.cwn].reshape(sa, plan.cwn) for j in range(plan.nprod)]
                out[k] = plan_up(plan, P)
            else:
                out[k] = np.zeros((0, plan.N // 64), dtype=np.uint64)

        def copy_local(dst, src):
            dst[...] = src
        return sharding.StrassenShardedStep(plan, rank, bufs, do_down, do_product, do_up, exchange, copy_local, chunks)

    sharding.run_products(make_step, len(pairs), inflight)
    return [out[k] for k in range(len(pairs))], sharding.local_rows(plan, rank, 0)


def assemble(plan, parts, m, n):
    """parts: {rank: (C_local, runs)} -> the m x n result as masked words."""
    wn = (n + 63) // 64
    C = np.zeros((plan.M, plan.N // 64), dtype=np.uint64)
    seen = np.zeros(plan.M, dtype=np.int32)
    for rank, (CL, runs) in parts.items():
        s = runs[0][1]
        for b, (g0, rows) in enumerate(runs):
            C[g0:g0 + rows] = CL[b * s:b * s + rows]
            seen[g0:g0 + rows] += 1
    assert (seen == 1).all(), "the ranks' slabs must tile the rows of C exactly once"
    assert not C[m:].any() and not C[:, wn:].any(), "padding rows / words of C must come out zero"
    return C[:m, :wn]
