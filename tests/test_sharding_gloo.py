"""The N > 1 path on CPU: m4ri_amd/sharding.py under torch.distributed with the gloo backend, two
processes.  The block products are done by the CPU oracle here (tests may use it as the checker's
stand-in for the device multiply); the partitioning, the pairwise XOR exchange and the ownership of the
reduced rows are exactly the code bench.py runs under RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from m4ri_amd import sharding  # noqa: E402
from m4ri_amd.mzd import Mzd  # noqa: E402


def _free_port():
    """A token for the rendezvous file name (the ranks meet through a file, not a TCP port)."""
    _free_port.n = getattr(_free_port, "n", 0) + 1
    return _free_port.n


def _worker(rank, world, port, grid, m, l, n, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    # rendezvous through a file in the test's own directory: no port to pick, nothing to collide with when suites run side by side
    dist.init_process_group("gloo", init_method=f"file://{os.path.join(out_dir, 'rendezvous_' + str(port))}", rank=rank, world_size=world)
    import cpu_libs
    orc = cpu_libs.oracle()
    A, B = Mzd.random(m, l, 3), Mzd.random(l, n, 4)  # every rank regenerates the operands, as bench.py does
    plan = sharding.make_plan(world, rank, m, l, n, grid=grid)
    state = {}

    def multiply(r0, r1, k0, k1, c0, c1):
        a = A.window(r0, k0, r1, k1).copy()
        b = B.window(k0, c0, k1, c1).copy()
        P = orc.mul(None, a, b, 0)
        state["P"] = torch.from_numpy(P.rows().view(np.int64).copy())  # (rows x stride) words

    def send_recv(partner, send_rows, recv_rows):
        P = state["P"]
        recv = torch.empty((recv_rows[1] - recv_rows[0], P.shape[1]), dtype=torch.int64)
        ops = [dist.P2POp(dist.isend, P[send_rows[0]:send_rows[1]].contiguous(), partner),
               dist.P2POp(dist.irecv, recv, partner)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        return recv

    def xor_rows(rows, got):
        state["P"][rows[0]:rows[1]] ^= got

    r0, r1, c0, c1 = sharding.run_sharded(plan, multiply, xor_rows, send_recv)
    b0, _ = plan.row_range()
    owned = state["P"][r0 - b0:r1 - b0].numpy().view(np.uint64)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), region=np.array([r0, r1, c0, c1]), words=owned)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("grid,m,l,n", [((2, 1, 1), 300, 200, 260),      # rows of C, no exchange
                                        ((1, 2, 1), 130, 257, 512),      # columns of C
                                        ((1, 1, 2), 200, 512, 200),      # inner dimension split + pairwise XOR exchange
                                        ((1, 1, 2), 77, 129, 65)])       # ragged: last slices take the remainders
def test_two_rank_sharded_product_matches_oracle(tmp_path, oracle, grid, m, l, n):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, grid, m, l, n, str(tmp_path)), nprocs=world, join=True)
    want = oracle.mul(None, Mzd.random(m, l, 3), Mzd.random(l, n, 4), 0)
    covered = np.zeros((m, want.width), dtype=bool)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        r0, r1, c0, c1 = (int(x) for x in z["region"])
        w0, w1 = c0 // 64, (c1 + 63) // 64
        got = z["words"][:, : w1 - w0]
        exp = want.masked()[r0:r1, w0:w1]
        assert np.array_equal(got, exp), (grid, r, (r0, r1, c0, c1))
        covered[r0:r1, w0:w1] = True
    assert covered.all(), "the ranks' reduced regions must tile C exactly"


def test_plans_tile_the_product():
    """Every default grid covers C exactly once and splits the inner dimension without gaps."""
    for world, grid in ((1, None), (2, None), (4, None), (8, None), (8, (2, 2, 2)), (4, (1, 2, 2))):
        m = l = n = 65536
        seen = {}
        for rank in range(world):
            p = sharding.make_plan(world, rank, m, l, n, grid=grid)
            r0, r1 = p.row_range(); c0, c1 = p.col_range(); k0, k1 = p.inner_range()
            assert c0 % 64 == 0 and k0 % 64 == 0
            seen.setdefault((r0, r1, c0, c1), []).append((k0, k1))
            o0, o1 = p.owned_rows_after_reduce()
            assert r0 <= o0 < o1 <= r1
        area = 0
        for (r0, r1, c0, c1), ks in seen.items():
            area += (r1 - r0) * (c1 - c0)
            ks.sort()
            assert ks[0][0] == 0 and ks[-1][1] == l and all(a[1] == b[0] for a, b in zip(ks, ks[1:]))
        assert area == m * n
    # the defaults never split the inner dimension: no exchange on the data path
    assert sharding.default_grid(8) == (4, 2, 1) and sharding.default_grid(4) == (2, 2, 1)


def _slab_worker(rank, world, port, m, l, n, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    # rendezvous through a file in the test's own directory: no port to pick, nothing to collide with when suites run side by side
    dist.init_process_group("gloo", init_method=f"file://{os.path.join(out_dir, 'rendezvous_' + str(port))}", rank=rank, world_size=world)
    import cpu_libs
    orc = cpu_libs.oracle()
    A, B = Mzd.random(m, l, 3), Mzd.random(l, n, 4)
    rc, bc = sharding.slab_cuts(m, world), sharding.slab_cuts(l, world)
    kb = sharding.slab_rows(l, world)
    mine_b = torch.zeros((kb, B.width), dtype=torch.int64)           # the short last slab is padded: equal pieces for the collective
    mine_b[:bc[rank + 1] - bc[rank]] = torch.from_numpy(B.masked()[bc[rank]:bc[rank + 1]].view(np.int64).copy())
    full_b = torch.empty((world * kb, B.width), dtype=torch.int64)
    sharding.all_gather_rows(dist, full_b, mine_b, staged=True)      # the variant's one collective (bench.py: RCCL all-gather)
    Bg = Mzd(l, n)
    Bg.valid_words()[:, :] = full_b[:l].numpy().view(np.uint64)
    if rc[rank + 1] > rc[rank]:
        As = A.window(rc[rank], 0, rc[rank + 1], l).copy()
        words = orc.mul(None, As, Bg, 0).masked()
    else:
        words = np.zeros((0, Bg.width if n % 64 == 0 else (n + 63) // 64), dtype=np.uint64)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rows=np.array([rc[rank], rc[rank + 1]]), words=words)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,m,l,n", [(2, 300, 256, 321), (2, 301, 257, 130),   # ragged: the last slab one row short
                                         (3, 100, 64, 70), (3, 4, 5, 64)])          # W does not divide anything; an EMPTY last slab of A
def test_row_slabs_with_all_gather(tmp_path, oracle, world, m, l, n):
    mp.spawn(_slab_worker, args=(world, _free_port(), m, l, n, str(tmp_path)), nprocs=world, join=True)
    want = oracle.mul(None, Mzd.random(m, l, 3), Mzd.random(l, n, 4), 0).masked()
    seen = 0
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        r0, r1 = (int(x) for x in z["rows"])
        assert np.array_equal(z["words"], want[r0:r1])
        seen += r1 - r0
    assert seen == m


def test_slab_cuts_and_default_variant():
    assert sharding.slab_cuts(65536, 8) == [8192 * k for k in range(9)]
    assert sharding.slab_cuts(10, 4) == [0, 3, 6, 9, 10] and sharding.slab_cuts(4, 3) == [0, 2, 4, 4] and sharding.slab_rows(4, 3) == 2
    for rows, world in ((1, 8), (7, 8), (100003, 8), (65537, 3)):
        c = sharding.slab_cuts(rows, world)
        assert c[0] == 0 and c[-1] == rows and all(a <= b for a, b in zip(c, c[1:])) and max(b - a for a, b in zip(c, c[1:])) == sharding.slab_rows(rows, world)
    assert sharding.default_variant(2) == "slabs" and sharding.default_variant(4) == "slabs" and sharding.default_variant(8) == "strassen"
    # shape-aware: BASELINE.json configs[3] shards the top Strassen level, configs[4] (l = 8192: nothing for a level to save) takes
    # row slabs + one all-gather of B at every world size, and so does any thin product
    assert sharding.default_variant(8, 65536, 65536, 65536) == "strassen"
    assert sharding.default_variant(8, 131072, 8192, 131072) == "slabs"
    assert sharding.default_variant(8, 4096, 65536, 65536) == "slabs" and sharding.default_variant(8, 65536, 65536, 4096) == "slabs"
    # 3 and 4 ranks: the 47 sub-products of the rank-47 scheme in batched products where they are at least 16384 on every side (measured
    # at 65536^3: 6.58 ms per rank against the row slab's 8.00), row slabs below that and on 2 ranks (one link)
    assert sharding.default_variant(4, 65536, 65536, 65536) == "strassen" and sharding.default_variant(3, 65536, 65536, 65536) == "strassen"
    assert sharding.default_variant(4, 32768, 32768, 32768) == "slabs" and sharding.default_variant(2, 65536, 65536, 65536) == "slabs"
    assert sharding.default_variant(4, 131072, 8192, 131072) == "slabs"


def test_slab_product_pieces_cover_the_inner_dimension_own_slab_first():
    """The overlapped row-slab product: C_r = A_r[:, own] * B_own first (nothing to wait for), then the pieces before and after the
    rank's own slab from the gathered B -- together the whole inner dimension, once."""
    for rows, world in ((65536, 2), (65536, 4), (8192, 8), (100, 3), (4, 3)):
        cuts = sharding.slab_cuts(rows, world)
        for rank in range(world):
            pieces = sharding.slab_product_pieces(cuts, rank)
            if cuts[rank + 1] > cuts[rank]:
                assert pieces[0] == (cuts[rank], cuts[rank + 1], True)
            assert all(not own for _, _, own in pieces[1:]) and sum(b - a for a, b, _ in pieces) == rows
            cover = sorted((a, b) for a, b, _ in pieces)
            assert cover[0][0] == 0 and cover[-1][1] == rows and all(x[1] == y[0] for x, y in zip(cover, cover[1:]))
    assert sharding.slab_product_pieces([0, 16384, 32768, 49152, 65536], 1) == [(16384, 32768, True), (0, 16384, False), (32768, 65536, False)]
