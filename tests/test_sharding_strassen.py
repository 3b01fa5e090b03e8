"""The Strassen-sharded multi-GPU product on CPU (include/m4ri_amd.h part 4, m4ri_amd/csrc/multi.hip,
m4ri_amd/sharding.py): the plan and piece table of the C library and the exchange walk of sharding.py,
with the device steps replaced by numpy/oracle stand-ins (tests/shard_sim.py).

  * every world size / level count in one process (ranks as threads over a mailbox),
  * world_size 2 under torch.distributed with the gloo backend -- exactly the transport code bench.py
    runs under RCCL."""
import os
import sys
import threading

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import m4ri_amd  # noqa: E402
import shard_sim  # noqa: E402
from m4ri_amd import sharding  # noqa: E402
from m4ri_amd.mzd import Mzd  # noqa: E402


def test_plan_arithmetic():
    for world, want in ((2, 2), (3, 2), (4, 2), (7, 1), (8, 2)):
        assert m4ri_amd.shard_plan(world, 65536, 65536, 65536).levels == want      # fewest rounds on the busiest rank
    assert m4ri_amd.shard_plan(4, 4096, 4096, 4096).levels == 1                    # second level only on large sub-products
    # 8 ranks: two levels are ONE application of the rank-47 scheme of the 4 x 4 x 4 block product (47 sub-products of (n/4)^3: 6 rounds,
    # 6/47 of the work on the busiest rank) -- ahead of one Winograd level (7 sub-products: 1/7) and of Strassen-Winograd twice (49: 7/49)
    p = m4ri_amd.shard_plan(8, 65536, 65536, 65536)
    assert (p.nprod, p.blocks, p.bm, p.bl, p.cwl, p.cwn) == (47, 4, 16384, 16384, 256, 256)
    assert [len(sharding.owned_products(p, r)) for r in range(8)] == [6] * 7 + [5]
    # every directed link of the mesh carries the same share of an operand: 1/8 of 32 MiB
    per_link = {}
    for side, j, r, pc in sharding.strassen_pieces(p, (0,)):
        if pc.holder != pc.owner:
            per_link[(pc.holder, pc.owner)] = per_link.get((pc.holder, pc.owner), 0) + pc.words * 8
    assert set(per_link.values()) == {6 * (4 << 20), 5 * (4 << 20)} and len(per_link) == 8 * 7
    # one level on request: the 7 sub-products of one Winograd level, one per rank
    p = m4ri_amd.shard_plan(8, 65536, 65536, 65536, 1)
    assert (p.nprod, p.blocks, p.bm, p.bl, p.cwl, p.cwn) == (7, 2, 32768, 32768, 512, 512)
    assert [len(sharding.owned_products(p, r)) for r in range(8)] == [1] * 7 + [0]
    per_link = {}
    for side, j, r, pc in sharding.strassen_pieces(p, (0,)):
        if pc.holder != pc.owner:
            per_link[(pc.holder, pc.owner)] = per_link.get((pc.holder, pc.owner), 0) + pc.words * 8
    assert set(per_link.values()) == {16 << 20} and len(per_link) == 7 * 7
    p = m4ri_amd.shard_plan(4, 65536, 65536, 65536)
    assert p.nprod == 47 and [len(sharding.owned_products(p, r)) for r in range(4)] == [12, 12, 12, 11]
    # slabs the scheme's passes do not take (children narrower than 64 words): Strassen-Winograd twice, 49
    p = m4ri_amd.shard_plan(4, 8192, 8192, 8192, 2)
    assert p.nprod == 49 and [len(sharding.owned_products(p, r)) for r in range(4)] == [13, 12, 12, 12]
    # ragged: padded to whole blocks / words, pieces tile every operand exactly once
    p = m4ri_amd.shard_plan(3, 1001, 777, 130, 2)
    assert (p.M, p.L, p.N) == (1004, 1024, 256) and p.bm == 251 and p.cwl == 4 and p.cwn == 1
    for side, rows, cw in ((0, p.bm, p.cwl), (1, p.bl, p.cwn), (2, p.bm, p.cwn)):
        for j in range(p.nprod):
            cover = np.zeros(rows * cw, dtype=np.int32)
            for r in range(p.world):
                pc = m4ri_amd.shard_piece(p, side, j, r)
                assert pc.owner == j % 3 and pc.holder == r
                base = (j // 3) * rows * cw
                cover[pc.owner_off - base:pc.owner_off - base + pc.words] += 1
            assert (cover == 1).all()


@pytest.mark.parametrize("chunks", [1, 2, 3, (2, 2), "3x2"])   # row (x column) chunks of the overlapped schedule: same batches on every rank, same bits
@pytest.mark.parametrize("world,levels,m,l,n", [(2, 1, 64, 128, 128), (2, 2, 100, 256, 256), (3, 1, 77, 130, 65),
                                                (4, 2, 203, 300, 257), (8, 1, 130, 129, 200), (8, 2, 64, 512, 256),
                                                (5, 2, 7, 64, 64)])   # more ranks than rows per block: empty slabs
def test_all_ranks_in_one_process(oracle, world, levels, m, l, n, chunks):
    _all_ranks_in_one_process(oracle, world, levels, m, l, n, chunks, 1)


@pytest.mark.parametrize("world,m,group", [(8, 100, 1), (8, 100, 2), (8, 64, 3), (8, 33, 6), (3, 50, 4), (5, 40, 2)])
def test_the_scheme_split_all_ranks_in_one_process(oracle, world, m, group):
    """Two sharded levels as ONE application of the rank-47 scheme of the 4 x 4 x 4 block product (47 sub-products; multi.hip: nprod_of) with
    the rounds of a rank multiplied `group` at a time (sharding.StrassenShardedStep: product_group): same batches on the links, same bits."""
    plan = m4ri_amd.shard_plan(world, m, 16384, 16384, 2)
    assert plan.nprod == 47 and plan.cwl == 64 and plan.cwn == 64
    _all_ranks_in_one_process(oracle, world, 2, m, 16384, 16384, 1, group)


def _all_ranks_in_one_process(oracle, world, levels, m, l, n, chunks, group):
    A, B = Mzd.random(m, l, 3), Mzd.random(l, n, 4)
    plan = m4ri_amd.shard_plan(world, m, l, n, levels)
    mail, lock, barrier = {}, threading.Lock(), threading.Barrier(world)
    parts, errors = {}, []

    def make_exchange(rank):
        def exchange(sends, recvs):
            with lock:
                for dst, v in sends:
                    mail.setdefault((rank, dst), []).append(np.array(v, copy=True))
            barrier.wait()
            for src, v in recvs:
                with lock:
                    v[:] = mail[(src, rank)].pop(0)
            barrier.wait()
        return exchange

    def work(rank):
        try:
            parts[rank] = shard_sim.rank_part(plan, rank, A, B, oracle, make_exchange(rank), chunks=chunks, group=group)
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    assert all(not q for q in mail.values()), "every posted piece must have been received"
    got = shard_sim.assemble(plan, parts, m, n)
    assert np.array_equal(got, oracle.mul(None, A, B, 0).masked())


def _free_port():
    """A token for the rendezvous file name (the ranks meet through a file, not a TCP port)."""
    _free_port.n = getattr(_free_port, "n", 0) + 1
    return _free_port.n


def _worker(rank, world, port, levels, m, l, n, out_dir, chunks, group=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    # rendezvous through a file in the test's own directory: no port to pick, nothing to collide with when suites run side by side
    dist.init_process_group("gloo", init_method=f"file://{os.path.join(out_dir, 'rendezvous_' + str(port))}", rank=rank, world_size=world)
    import cpu_libs
    orc = cpu_libs.oracle()
    A, B = Mzd.random(m, l, 3), Mzd.random(l, n, 4)
    plan = m4ri_amd.shard_plan(world, m, l, n, levels)
    torch_x = sharding.torch_exchange(dist)

    def exchange(sends, recvs):  # numpy views <-> torch tensors sharing memory: the transport is sharding.torch_exchange
        torch_x([(d, torch.from_numpy(v.view(np.int64))) for d, v in sends], [(s, torch.from_numpy(v.view(np.int64))) for s, v in recvs])

    exchange.post = lambda sends, recvs: torch_x.post([(d, torch.from_numpy(v.view(np.int64))) for d, v in sends],
                                                      [(s, torch.from_numpy(v.view(np.int64))) for s, v in recvs])
    CL, runs = shard_sim.rank_part(plan, rank, A, B, orc, exchange, chunks=chunks, group=group)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), C=CL, runs=np.array(runs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("levels,m,l,n,chunks", [(1, 200, 256, 320, 1), (2, 131, 257, 129, 1), (1, 200, 256, 320, 2), (2, 131, 257, 129, 2),
                                                 (1, 200, 256, 320, "2x2"), (2, 131, 257, 513, "2x3")])
def test_two_ranks_gloo(tmp_path, oracle, levels, m, l, n, chunks):
    _two_ranks_gloo(tmp_path, oracle, levels, m, l, n, chunks, 1)


@pytest.mark.parametrize("group", [1, 4])
def test_the_scheme_split_two_ranks_gloo(tmp_path, oracle, group):
    """The 47-way split over two gloo ranks, the rank's 24 / 23 sub-products one at a time and four at a time."""
    assert m4ri_amd.shard_plan(2, 70, 16384, 16384, 2).nprod == 47
    _two_ranks_gloo(tmp_path, oracle, 2, 70, 16384, 16384, 1, group)


def _two_ranks_gloo(tmp_path, oracle, levels, m, l, n, chunks, group):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), levels, m, l, n, str(tmp_path), chunks, group), nprocs=world, join=True)
    plan = m4ri_amd.shard_plan(world, m, l, n, levels)
    parts = {}
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        parts[r] = (z["C"], [tuple(int(x) for x in row) for row in z["runs"]])
    got = shard_sim.assemble(plan, parts, m, n)
    assert np.array_equal(got, oracle.mul(None, Mzd.random(m, l, 3), Mzd.random(l, n, 4), 0).masked())


def test_chunk_bounds_cover_the_rows_once():
    """The overlapped schedule's row chunks are whole slabs, tile the sub-product's rows and never cut a piece."""
    for world, m, chunks in ((8, 65536, 2), (8, 65536, 4), (8, 65536, 3), (4, 65536, 2), (3, 1001, 2), (5, 7, 2), (2, 64, 1)):
        plan = m4ri_amd.shard_plan(world, m, 4096, 4096)
        b = sharding.chunk_bounds(plan, chunks)
        assert b[0][2] == 0 and sum(x[3] for x in b) == plan.bm and all(x[2] + x[3] == y[2] for x, y in zip(b, b[1:]))
        assert b[0][0] == 0 and b[-1][1] == world and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    plan = m4ri_amd.shard_plan(8, 65536, 65536, 65536, 1)
    assert [x[2:] for x in sharding.chunk_bounds(plan, 2)] == [(0, 16384), (16384, 16384)]   # halves of a 32768-row sub-product
    assert sharding.column_bounds(plan, 2) == [(0, 256), (256, 512)] and sharding.column_bounds(plan, 1) == [(0, 512)]
    assert sharding.column_bounds(m4ri_amd.shard_plan(2, 64, 64, 64), 4) == [(0, 1)]        # one word per row: nothing to cut
    assert sharding.parse_chunks("2x2") == (2, 2) and sharding.parse_chunks(3) == (3, 1) and sharding.parse_chunks("2") == (2, 1)


@pytest.mark.parametrize("world,levels,chunks,inflight", [(8, 1, 2, 2), (4, 2, 1, 2), (3, 1, "2x2", 2), (2, 2, 1, 1)])
def test_products_in_flight_on_two_buffer_slots(oracle, world, levels, chunks, inflight):
    """sharding.run_products: start(k+1), multiply(k), finish(k-1) over two buffer slots, five DIFFERENT products in a row -- every
    C_k must be its own product (a slot reused too early would mix neighbours)."""
    m, l, n = 130, 257, 200
    pairs = [(Mzd.random(m, l, 100 + k), Mzd.random(l, n, 200 + k)) for k in range(5)]
    plan = m4ri_amd.shard_plan(world, m, l, n, levels)
    mail, lock, barrier = {}, threading.Lock(), threading.Barrier(world)
    parts, errors = {}, []

    def make_exchange(rank):
        def exchange(sends, recvs):
            with lock:
                for dst, v in sends:
                    mail.setdefault((rank, dst), []).append(np.array(v, copy=True))
            barrier.wait()
            for src, v in recvs:
                with lock:
                    v[...] = mail[(src, rank)].pop(0)
            barrier.wait()
        return exchange

    def work(rank):
        try:
            parts[rank] = shard_sim.rank_products(plan, rank, pairs, oracle, make_exchange(rank), inflight, chunks)
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for k, (A, B) in enumerate(pairs):
        got = shard_sim.assemble(plan, {r: (parts[r][0][k], parts[r][1]) for r in range(world)}, m, n)
        assert np.array_equal(got, oracle.mul(None, A, B, 0).masked()), k


def _pipeline_worker(rank, world, port, levels, m, l, n, out_dir, chunks):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    # rendezvous through a file in the test's own directory: no port to pick, nothing to collide with when suites run side by side
    dist.init_process_group("gloo", init_method=f"file://{os.path.join(out_dir, 'rendezvous_' + str(port))}", rank=rank, world_size=world)
    import cpu_libs
    orc = cpu_libs.oracle()
    pairs = [(Mzd.random(m, l, 300 + k), Mzd.random(l, n, 400 + k)) for k in range(4)]
    plan = m4ri_amd.shard_plan(world, m, l, n, levels)
    torch_x = sharding.torch_exchange(dist)   # CPU tensors under gloo: isend / irecv really run in the background until wait()

    def as_t(v):
        return torch.from_numpy(v.view(np.int64))

    def exchange(sends, recvs):
        torch_x([(d, as_t(v)) for d, v in sends], [(s_, as_t(v)) for s_, v in recvs])
    exchange.post = lambda sends, recvs: torch_x.post([(d, as_t(v)) for d, v in sends], [(s_, as_t(v)) for s_, v in recvs])
    Cs, runs = shard_sim.rank_products(plan, rank, pairs, orc, exchange, 2, chunks)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), runs=np.array(runs), **{f"C{k}": c for k, c in enumerate(Cs)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("levels,chunks", [(1, 2), (2, 1)])
def test_two_products_in_flight_two_ranks_gloo(tmp_path, oracle, levels, chunks):
    """run_products over the asynchronous transport (torch_exchange.post: batches posted, waited for later) with two ranks under gloo:
    four different products through two buffer slots, each against the oracle."""
    world, (m, l, n) = 2, (200, 256, 320)
    mp.spawn(_pipeline_worker, args=(world, _free_port(), levels, m, l, n, str(tmp_path), chunks), nprocs=world, join=True)
    plan = m4ri_amd.shard_plan(world, m, l, n, levels)
    z = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    for k in range(4):
        parts = {r: (z[r][f"C{k}"], [tuple(int(x) for x in row) for row in z[r]["runs"]]) for r in range(world)}
        got = shard_sim.assemble(plan, parts, m, n)
        assert np.array_equal(got, oracle.mul(None, Mzd.random(m, l, 300 + k), Mzd.random(l, n, 400 + k), 0).masked()), k
