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
