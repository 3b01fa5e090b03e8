"""Multi-GPU decomposition of one product C = A*B: one process per GPU, block products sharded with
no data-path collective except ONE pairwise XOR exchange when the inner dimension is split.

The template is the reference's own multi-core path, _mzd_mul_mp4 (reference m4ri/mp.c:158-275): C is
split into blocks and every rank computes its block(s) C_ij = A_i* x B_*j.  The default grids split
only the rows and columns of C, so no rank ever needs another rank's data:

    world 1: (1,1,1)   world 2: (2,1,1) rows of C   world 4: (2,2,1)   world 8: (4,2,1) blocks of C

(the engine gives the resulting rectangular blocks, e.g. 16384 x 65536 x 32768 at 8 GPUs, the same
Strassen depth per leaf size as a cube).  A grid may also split the inner dimension (gh > 1, e.g.
(2,2,2)): the gh partial products of a block then sit on different GPUs and are combined by a
pairwise exchange over xGMI; RCCL has no XOR reduction, so the reduce is "send/recv half of the
partial product + local XOR kernel" -- each pair talks over its own point-to-point link, nothing is
ring-shaped.  That costs 64 MiB per rank and product at n = 65536, which is why it is not the default.

Everything here is device-agnostic (views are (row0, rows, col0_bits, cols_bits) tuples; the multiply,
XOR and transport are injected), so the same code runs under gloo on CPU tensors in the tests and
under RCCL on HBM tensors in bench.py.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class ShardPlan:
    world: int
    rank: int
    grid: tuple  # (gi, gj, gh): splits of m, n and of the inner dimension l
    i: int
    j: int
    h: int
    m: int
    l: int
    n: int

    # ---- geometry: splits land on multiples of `align` (rows) / 64*align (bit columns) ------
    @staticmethod
    def _cuts(total: int, parts: int, unit: int):
        base = (total // parts) // unit * unit
        cuts = [k * base for k in range(parts)] + [total]
        if parts > 1 and base == 0:
            cuts = [0] * parts + [total]  # degenerate: last part takes everything
        return cuts

    def row_range(self):
        c = self._cuts(self.m, self.grid[0], 1)
        return c[self.i], c[self.i + 1]

    def col_range(self):
        c = self._cuts(self.n, self.grid[1], 64)
        return c[self.j], c[self.j + 1]

    def inner_range(self):
        c = self._cuts(self.l, self.grid[2], 64)
        return c[self.h], c[self.h + 1]

    def partner_ranks(self):
        """Ranks holding the other inner-dimension slices of the same C block (h != self.h)."""
        gi, gj, gh = self.grid
        return [rank_of((self.i, self.j, hh), self.grid) for hh in range(gh) if hh != self.h]

    def owned_rows_after_reduce(self):
        """After the exchange rank h keeps row slice h (of gh) of the block's rows, fully reduced."""
        r0, r1 = self.row_range()
        c = self._cuts(r1 - r0, self.grid[2], 1)
        return r0 + c[self.h], r0 + c[self.h + 1]


def default_grid(world: int):
    return {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (4, 2, 1)}.get(world) or _fallback_grid(world)


def _fallback_grid(world: int):
    # rows only: always valid, never needs an exchange
    return (world, 1, 1)


def rank_of(ijh, grid):
    i, j, h = ijh
    return (i * grid[1] + j) * grid[2] + h


def make_plan(world: int, rank: int, m: int, l: int, n: int, grid=None) -> ShardPlan:
    grid = tuple(grid) if grid is not None else default_grid(world)
    assert grid[0] * grid[1] * grid[2] == world, (grid, world)
    h = rank % grid[2]
    j = (rank // grid[2]) % grid[1]
    i = rank // (grid[2] * grid[1])
    return ShardPlan(world, rank, grid, i, j, h, m, l, n)


def run_sharded(plan: ShardPlan, multiply, xor_rows, send_recv):
    """One sharded product.

    multiply(r0, r1, k0, k1, c0, c1) -> handle of the local partial P = A[r0:r1, k0:k1] * B[k0:k1, c0:c1]
    send_recv(partner_rank, send_rows, recv_rows) -> handle of the received rows: ships rows
        [send_rows) of the local P (block-relative) to the partner and receives the partner's rows
        [recv_rows) of ITS P;
    xor_rows(rows, received) : P[rows] ^= received.

    Returns (r0, r1, c0, c1): the region of C this rank holds fully reduced inside its P afterwards.
    """
    r0, r1 = plan.row_range()
    c0, c1 = plan.col_range()
    k0, k1 = plan.inner_range()
    multiply(r0, r1, k0, k1, c0, c1)
    gh = plan.grid[2]
    if gh == 1:
        return r0, r1, c0, c1
    # pairwise XOR exchange: I keep row slice `h` of the block, every partner sends me its copy of
    # that slice and gets from me the slice it keeps
    cuts = ShardPlan._cuts(r1 - r0, gh, 1)
    mine = (cuts[plan.h], cuts[plan.h + 1])
    for pr in plan.partner_ranks():
        ph = pr % gh
        theirs = (cuts[ph], cuts[ph + 1])
        got = send_recv(pr, theirs, mine)
        xor_rows(mine, got)
    return r0 + mine[0], r0 + mine[1], c0, c1


# ==================================================================================================
# Strassen-level sharding (include/m4ri_amd.h part 4, m4ri_amd/csrc/multi.hip): the sub-products of the
# top Strassen-Winograd level(s) over the ranks, matrices distributed slab-cyclically.
#
# One process per GPU.  The plan and the piece table come from the C library (pure host arithmetic);
# this module only walks the table and hands every piece to the injected transport, so the very same
# code runs under RCCL on HBM tensors (bench.py), under gloo on one shared GPU (tests -m gpu) and
# under gloo on CPU tensors (tests/test_sharding_gloo.py).
# ==================================================================================================
def strassen_pieces(plan, sides=(0, 1)):
    """All pieces (side, j, r) of the plan in the canonical order both ends of a link post them in."""
    import m4ri_amd
    out = []
    for side in sides:
        for j in range(plan.nprod):
            for r in range(plan.world):
                pc = m4ri_amd.shard_piece(plan, side, j, r)
                if pc.words:
                    out.append((side, j, r, pc))
    return out


def owned_products(plan, rank):
    return list(range(rank, plan.nprod, plan.world))


def run_strassen_sharded(plan, rank, bufs, down, product, up, exchange, copy_local):
    """One product C = A*B over plan.world ranks; this is rank `rank`'s part.

    bufs: dict of 1-D word tensors/arrays keyed 'child_a', 'child_b', 'slabs_p', 'oper_a', 'oper_b', 'prod'
          (sizes: m4ri_amd.shard_buffer_words).
    down():            local parents of A and B -> bufs['child_a'], bufs['child_b']   (local Winograd down pass)
    product(jl, j):    bufs['prod'][jl] = bufs['oper_a'][jl] * bufs['oper_b'][jl]      (owned sub-product number jl)
    up():              bufs['slabs_p'] -> local parent of C                            (local Winograd up pass)
    exchange(sends, recvs): sends = [(dst_rank, view)], recvs = [(src_rank, view)], each list in the
                       canonical piece order; must complete before returning.
    copy_local(dst_view, src_view): a piece whose holder and owner are this rank.
    """
    child = {0: bufs["child_a"], 1: bufs["child_b"]}
    oper = {0: bufs["oper_a"], 1: bufs["oper_b"]}
    down()
    sends, recvs = [], []
    for side, j, r, pc in strassen_pieces(plan, (0, 1)):
        src = child[side][pc.holder_off:pc.holder_off + pc.words] if pc.holder == rank else None
        dst = oper[side][pc.owner_off:pc.owner_off + pc.words] if pc.owner == rank else None
        if pc.holder == rank and pc.owner == rank:
            copy_local(dst, src)
        elif pc.holder == rank:
            sends.append((pc.owner, src))
        elif pc.owner == rank:
            recvs.append((pc.holder, dst))
    exchange(sends, recvs)
    for jl, j in enumerate(owned_products(plan, rank)):
        product(jl, j)
    sends, recvs = [], []
    for side, j, r, pc in strassen_pieces(plan, (2,)):
        src = bufs["prod"][pc.owner_off:pc.owner_off + pc.words] if pc.owner == rank else None
        dst = bufs["slabs_p"][pc.holder_off:pc.holder_off + pc.words] if pc.holder == rank else None
        if pc.holder == rank and pc.owner == rank:
            copy_local(dst, src)
        elif pc.owner == rank:
            sends.append((pc.holder, src))
        elif pc.holder == rank:
            recvs.append((pc.owner, dst))
    exchange(sends, recvs)
    up()


def local_rows(plan, rank, which):
    """Global row indices (of A/C for which == 0, of B for which == 1) held by `rank`, in local order:
    slab `rank` of block 0, of block 1, ... -- as (global_row0, rows) runs."""
    import m4ri_amd
    brows = plan.bl if which else plan.bm
    c0 = int(m4ri_amd.lib().m4ri_amd_shard_cut(brows, plan.world, rank))
    c1 = int(m4ri_amd.lib().m4ri_amd_shard_cut(brows, plan.world, rank + 1))
    return [(b * brows + c0, c1 - c0) for b in range(plan.blocks)]


def torch_exchange(dist, staged_device=None):
    """Transport over torch.distributed P2P ops (backend nccl == RCCL: every piece is one send/recv pair on
    the direct xGMI link between its two ranks; all pieces of a phase are posted as one batch, so all links
    of the mesh work concurrently).  staged_device: tensors live on that device but the backend (gloo)
    cannot move them -- stage through the host (development / one-GPU tests only)."""
    import torch

    def exchange(sends, recvs):
        if not sends and not recvs:
            return
        if staged_device is not None:
            outs = [(dst, v.cpu()) for dst, v in sends]
            ins = [(src, torch.empty(v.shape, dtype=v.dtype), v) for src, v in recvs]
            ops = [dist.P2POp(dist.isend, t, dst) for dst, t in outs] + [dist.P2POp(dist.irecv, t, src) for src, t, _ in ins]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            for _, t, v in ins:
                v.copy_(t)
            return
        # RCCL moves contiguous tensors only: the pieces are contiguous by construction (row slabs, whole buffers), but a view
        # that is not goes through a contiguous temporary rather than aborting the job
        outs = [(dst, v if v.is_contiguous() else v.contiguous()) for dst, v in sends]
        ins = [(src, v if v.is_contiguous() else torch.empty(v.shape, dtype=v.dtype, device=v.device), v) for src, v in recvs]
        ops = [dist.P2POp(dist.isend, t, dst) for dst, t in outs] + [dist.P2POp(dist.irecv, t, src) for src, t, _ in ins]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for _, t, v in ins:
            if t is not v:
                v.copy_(t)
    return exchange


# ==================================================================================================
# Row slabs: the simplest distributed product, C_r = A_r * B with B replicated by ONE all-gather.
# Rank r holds rows [cut(r), cut(r+1)) of A, of B and (afterwards) of C -- the layout closes, products chain.
# The only communication is the all-gather of B's row slabs (an RCCL collective; on the xGMI mesh every link
# carries 1/W of B).  Each rank then multiplies its slab by the whole B with the full single-GPU engine; what the
# variant gives up is Strassen depth in the row direction (a slab of m/W rows has log2(W) levels fewer), which is
# why the Strassen-sharded variant overtakes it at 8 GPUs while the slabs win at 2 and 4.
# ==================================================================================================
def slab_cuts(rows: int, world: int):
    """Row boundaries of the W slabs (equal slabs; the all-gather wants equal pieces: rows must divide)."""
    assert rows % world == 0, (rows, world)
    return [k * (rows // world) for k in range(world + 1)]


def all_gather_rows(dist, full, mine, staged=False):
    """full (W * k rows) <- the ranks' `mine` (k rows each) in rank order: dist.all_gather_into_tensor, or its
    host-staged equivalent for backends that cannot move device tensors (gloo in the one-GPU tests)."""
    import torch
    if not staged:
        dist.all_gather_into_tensor(full.view(-1), mine.contiguous().view(-1))
        return
    parts = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine.cpu() if mine.is_cuda else mine.contiguous())
    k = mine.shape[0]
    for r, t in enumerate(parts):
        full[r * k:(r + 1) * k].copy_(t)


def default_variant(world: int) -> str:
    """slabs up to 4 ranks (one all-gather of B; 2 ranks share a single link, which the Strassen-sharded exchange
    would saturate), the Strassen sub-products from 5 ranks on (arithmetic for n = 65536)."""
    return "slabs" if world <= 4 else "strassen"
