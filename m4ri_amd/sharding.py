"""One product C = A*B over several GPUs, one process per GPU (bench.py drives it).

Three ways to hand the work out, all bit-identical to the single-GPU product:

  * Strassen sub-products (the design for 5+ ranks; StrassenShardedStep, run_strassen_sharded, run_products): the 7, 47 or 49
    sub-products of the top Strassen level(s) (reference m4ri/strassen.c:111-150; 47 = the rank-47 scheme of the 4 x 4 x 4 block
    product applied once to two levels, the choice is the C library's: plan.nprod) go to the ranks, a rank's own several in groups
    as batched products (group / product_group), the matrices are
    distributed slab-cyclically, so the level's additions are local passes and only slabs of operands / products cross the
    xGMI mesh -- every rank to every rank, one send/recv pair per piece, posted in batches laid out so that transport runs under
    the multiplications (row / column chunks inside one product, and -- for a stream of products -- two products in flight).
    The plan and the piece table come from libm4ri_amd.so (pure host arithmetic, include/m4ri_amd.h part 4).
  * Row slabs (2 and 4 ranks, and any product a Strassen level cannot help: default_variant): rank r holds rows of A, B and C;
    ONE collective, the all-gather of B (all_gather_rows), which runs under the product with the rank's own slab
    (slab_product_pieces) or under the previous product.  The reference's own row parallelism, m4ri/brilliantrussian.c:1121-1123.
  * Blocks of C (ShardPlan, run_sharded; kept for comparison): the reference's own multi-core template, _mzd_mul_mp4
    (m4ri/mp.c:158-275) -- every rank computes C_ij = A_i* x B_*j; a grid may also split the inner dimension (gh > 1), the gh
    partial products then meet by ONE pairwise exchange + a local XOR kernel (RCCL has no XOR reduction and none is needed).

Everything here is device-agnostic: the multiply, the local passes and the transport are injected, so the same walks run under
gloo on CPU arrays in the tests (tests/shard_sim.py), under gloo with several ranks sharing one GPU, and under RCCL on HBM
tensors in bench.py.  torch_exchange is the transport over torch.distributed.
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


class _Done:
    """Handle of an exchange that has already completed (synchronous transports)."""
    def wait(self):
        return None


def _post(exchange, sends, recvs):
    """Start one batch on the transport and return its handle (.wait()).  A transport with a `post` attribute moves the
    batch asynchronously (torch_exchange under RCCL: the batch runs on the communicator's stream, wait() makes the compute
    stream wait for it, the host never blocks); a plain callable completes before returning."""
    post = getattr(exchange, "post", None)
    if post is not None:
        return post(sends, recvs)
    exchange(sends, recvs)
    return _Done()


def chunk_bounds(plan, chunks: int):
    """Row chunks of a sub-product for the overlapped schedule: chunk c = the slabs of ranks [W c / chunks, W (c+1) / chunks),
    i.e. rows [cut(r_lo), cut(r_hi)) of the sub-product's A operand and of its result -- whole pieces, nothing is cut.
    Returns [(r_lo, r_hi, row0, rows)], empty chunks dropped."""
    import m4ri_amd
    out = []
    for c in range(max(1, chunks)):
        lo, hi = plan.world * c // max(1, chunks), plan.world * (c + 1) // max(1, chunks)
        r0 = int(m4ri_amd.lib().m4ri_amd_shard_cut(plan.bm, plan.world, lo))
        r1 = int(m4ri_amd.lib().m4ri_amd_shard_cut(plan.bm, plan.world, hi))
        if hi > lo:
            out.append((lo, hi, r0, r1 - r0))
    return out


def column_bounds(plan, col_chunks: int):
    """Column chunks of a sub-product's B operand and result, in whole words: [(w0, w1)] over the plan.cwn words of a row."""
    cc = max(1, min(col_chunks, plan.cwn))
    cuts = [plan.cwn * h // cc for h in range(cc + 1)]
    return [(a, b) for a, b in zip(cuts, cuts[1:]) if b > a]


def parse_chunks(spec):
    """'2' -> (2, 1); '2x2' -> (2, 2); an int or a pair passes through."""
    if isinstance(spec, (tuple, list)):
        return int(spec[0]), int(spec[1])
    if isinstance(spec, int):
        return spec, 1
    a, _, b = str(spec).lower().partition("x")
    return int(a), int(b) if b else 1


def _cols(view, row_words, w0, w1):
    """Words [w0, w1) of every row of a piece stored as whole rows of `row_words` words (a 2-D strided view; torch and numpy alike)."""
    if w0 == 0 and w1 == row_words:
        return view
    return view.reshape(-1, row_words)[:, w0:w1]


class StrassenShardedStep:
    """One sharded product as three phases, so that a caller with several products to multiply can keep more than one in flight
    (bench.py --inflight 2: start(k+1), multiply(k), finish(k-1) -- the operands of product k+1 and the results of product k-1
    cross the links while product k is multiplied).  start(); multiply(); finish() in a row is run_strassen_sharded.
    Arguments and schedule: run_strassen_sharded."""

    def __init__(self, plan, rank, bufs, down, product, up, exchange, copy_local, chunks=1, group=1, product_group=None):
        self.plan, self.rank, self.bufs = plan, rank, bufs
        self.down, self.product, self.up, self.exchange, self.copy_local = down, product, up, exchange, copy_local
        # group > 1 (whole sub-products only): rounds q0 .. q0 + group - 1 are multiplied by ONE call product_group(q0, count) -- a batched
        # product over the rank's consecutive operand / result buffers (m4ri_amd.mul_batch_dev); the batches on the links are the same
        self.group, self.product_group = (int(group), product_group) if product_group is not None else (1, None)
        self.W = plan.world
        self.rounds = -(-plan.nprod // self.W)
        rchunks, cchunks = parse_chunks(chunks)
        self.bounds = chunk_bounds(plan, rchunks)
        self.cbounds = column_bounds(plan, cchunks)
        self.table = {}
        for side, j, r, pc in strassen_pieces(plan, (0, 1, 2)):
            self.table.setdefault((side, j // self.W), []).append((r, pc))
        self.inbound, self.returns = {}, []

    def _pieces(self, side, q, r_lo, r_hi, w0, w1, sends, recvs):
        """The sends and receives of the pieces (side, round q, slabs r_lo .. r_hi, words w0 .. w1) appended to the two lists, in the
        canonical order; local pieces are copied at once."""
        plan, rank, bufs = self.plan, self.rank, self.bufs
        child = {0: bufs["child_a"], 1: bufs["child_b"]}
        oper = {0: bufs["oper_a"], 1: bufs["oper_b"]}
        row_words = plan.cwl if side == 0 else plan.cwn
        w1 = row_words if w1 is None else w1
        for r, pc in self.table.get((side, q), ()):
            if not (r_lo <= r < r_hi):
                continue
            if side < 2:   # operand slab: holder -> owner
                src = _cols(child[side][pc.holder_off:pc.holder_off + pc.words], row_words, w0, w1) if pc.holder == rank else None
                dst = _cols(oper[side][pc.owner_off:pc.owner_off + pc.words], row_words, w0, w1) if pc.owner == rank else None
                frm, to = pc.holder, pc.owner
            else:          # product slab: owner -> holder
                src = _cols(bufs["prod"][pc.owner_off:pc.owner_off + pc.words], row_words, w0, w1) if pc.owner == rank else None
                dst = _cols(bufs["slabs_p"][pc.holder_off:pc.holder_off + pc.words], row_words, w0, w1) if pc.holder == rank else None
                frm, to = pc.owner, pc.holder
            if frm == rank and to == rank:
                self.copy_local(dst, src)
            elif frm == rank:
                sends.append((to, src))
            elif to == rank:
                recvs.append((frm, dst))

    def _batch(self, side, q, r_lo, r_hi, w0=0, w1=None):
        sends, recvs = [], []
        self._pieces(side, q, r_lo, r_hi, w0, w1, sends, recvs)
        return _post(self.exchange, sends, recvs)

    def _grouped(self):
        return self.group > 1 and len(self.bounds) == 1 and len(self.cbounds) == 1

    def _group_batch(self, sides, q0):
        """ONE batch for the pieces of `sides` of every round of the group that starts at round q0 (whole sub-products): a group's
        operands -- and its results -- cross the links as one group of point-to-point transfers, so a product of 47 sub-products costs the
        host as many batches as one of 7 in two row chunks."""
        sends, recvs = [], []
        for q in range(q0, min(q0 + self.group, self.rounds)):
            for side in sides:
                self._pieces(side, q, 0, self.W, 0, None, sends, recvs)
        return _post(self.exchange, sends, recvs)

    def start(self):
        """Local down pass, then every outbound operand chunk posted, in the order the units will want them."""
        self.down()
        self.inbound = {}
        if self._grouped():
            for q0 in range(0, self.rounds, self.group):
                self.inbound[("group", q0)] = self._group_batch((1, 0), q0)
            return
        for q in range(self.rounds):
            for c, (lo, hi, _, _) in enumerate(self.bounds):
                for h, (w0, w1) in enumerate(self.cbounds):
                    if (1, q, h) not in self.inbound:
                        self.inbound[(1, q, h)] = self._batch(1, q, 0, self.W, w0, w1)
                    if (0, q, c) not in self.inbound:
                        self.inbound[(0, q, c)] = self._batch(0, q, lo, hi)

    def multiply(self):
        """Every unit: wait for its operand chunks, multiply, post its part of the result."""
        owned = owned_products(self.plan, self.rank)
        self.returns = []
        if self._grouped():
            for q0 in range(0, self.rounds, self.group):
                qs = range(q0, min(q0 + self.group, self.rounds))
                self.inbound[("group", q0)].wait()
                count = sum(1 for q in qs if q < len(owned))
                if count:
                    self.product_group(q0, count)
                self.returns.append(self._group_batch((2,), q0))
            return
        for q in range(self.rounds):
            for c, (lo, hi, row0, rows) in enumerate(self.bounds):
                for h, (w0, w1) in enumerate(self.cbounds):
                    self.inbound[(1, q, h)].wait()
                    self.inbound[(0, q, c)].wait()
                    if q < len(owned) and rows:
                        self.product(q, owned[q], row0, rows, w0, w1)
                    self.returns.append(self._batch(2, q, lo, hi, w0, w1))

    def finish(self):
        """Wait for the slabs of every product, then the local up pass."""
        for hnd in self.returns:
            hnd.wait()
        self.returns = []
        self.up()


def run_strassen_sharded(plan, rank, bufs, down, product, up, exchange, copy_local, chunks=1, group=1, product_group=None):
    """One product C = A*B over plan.world ranks; this is rank `rank`'s part.

    bufs: dict of 1-D word tensors/arrays keyed 'child_a', 'child_b', 'slabs_p', 'oper_a', 'oper_b', 'prod'
          (sizes: m4ri_amd.shard_buffer_words).
    down():            local parents of A and B -> bufs['child_a'], bufs['child_b']   (local Winograd down pass)
    product(jl, j, row0, rows, w0, w1): rows [row0, row0 + rows), words [w0, w1) of bufs['prod'][jl] = the same rows of
                       bufs['oper_a'][jl] times the same word columns of bufs['oper_b'][jl]   (owned sub-product number jl)
    up():              bufs['slabs_p'] -> local parent of C                            (local Winograd up pass)
    exchange:          the transport: exchange(sends, recvs) with sends = [(dst_rank, view)], recvs = [(src_rank, view)],
                       each list in the canonical piece order; optionally exchange.post(sends, recvs) -> handle with
                       .wait() for transports that run a batch in the background (see _post).
    copy_local(dst_view, src_view): a piece whose holder and owner are this rank.
    chunks:            row chunks, or (row chunks, column chunks) / 'RxC', per sub-product (see below).

    The schedule (the reference's block-parallel template has no transport to hide, m4ri/mp.c:191-228; here the pieces
    cross xGMI links, so the walk is laid out for overlap).  Sub-products are multiplied in ROUNDS (round q = the q-th
    owned product of every rank, j in [q W, (q+1) W)), each as R x C units: row chunk c (chunk_bounds: whole slabs) times
    column chunk h (column_bounds: whole words).  Unit (c, h) needs rows c of the A operand and columns h of the B operand
    and nothing else, and its part of the result is needed nowhere before the up pass.  All ranks post the same sequence of
    batches, every batch one group of point-to-point transfers:

        down;  for every round q, unit (c, h) in row-major order:  post B(q, h), A(q, c) unless already posted  <- all outbound, at once
        for every round q, unit (c, h):  wait B(q, h), A(q, c);  product of the unit;  post P(q, c, h)
        wait every P;  up

    so every operand chunk but the first unit's travels while an earlier unit is multiplied, and every result chunk but the
    last travels back under a later unit: exposed on the links are B(0, 0), A(0, 0) and the last P only.  chunks = 1 with a
    synchronous transport is the plain three-phase walk (same bits in every case).  Column chunks are strided views of the
    row-slab pieces; the transport packs what it cannot move as it is (torch_exchange: a contiguous temporary).
    """
    step = StrassenShardedStep(plan, rank, bufs, down, product, up, exchange, copy_local, chunks, group, product_group)
    step.start()
    step.multiply()
    step.finish()


def run_products(make_step, n, inflight=1, before=None):
    """n products, each an object with start() / multiply() / finish() (StrassenShardedStep, or anything shaped like it), made by
    make_step(k).  inflight = 1: one after the other.  inflight = 2: software-pipelined --

        start(0);   for k:  start(k+1);  multiply(k);  finish(k-1);      finish(n-1)

    so the transport a start() and a multiply() post (operands of product k+1, results of product k) runs under the multiplications of
    the neighbouring product.  Product k may reuse the buffers of product k-2 (two slots): start(k+1) touches the slot's child /
    operand buffers, which multiply(k-1) has finished with, and finish(k-1) the slot's slab / result buffers, which nothing writes
    before multiply(k+1).  before(k) is called ahead of product k's first action inside the loop (timing marks)."""
    if n <= 0:
        return
    if inflight <= 1:
        for k in range(n):
            if before is not None:
                before(k)
            st = make_step(k)
            st.start()
            st.multiply()
            st.finish()
        return
    assert inflight == 2, "one or two products in flight"
    cur, prev = make_step(0), None
    cur.start()
    for k in range(n):
        if before is not None:
            before(k)
        nxt = None
        if k + 1 < n:
            nxt = make_step(k + 1)
            nxt.start()
        cur.multiply()
        if prev is not None:
            prev.finish()
        prev, cur = cur, nxt
    prev.finish()


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
    the direct xGMI link between its two ranks; all pieces of a batch are posted as one group, so all links
    of the mesh work concurrently).  staged_device: tensors live on that device but the backend (gloo)
    cannot move them -- stage through the host (development / one-GPU tests only).

    The returned callable completes a batch before returning; its `.post(sends, recvs)` starts the batch and returns a
    handle whose wait() orders the CURRENT stream behind it (RCCL: the host does not block, so transfers overlap whatever
    the compute stream is doing meanwhile; gloo / staged: the batch is complete when post returns)."""
    import torch

    class _Pending:
        def __init__(self, reqs, copies):
            self.reqs, self.copies = reqs, copies

        def wait(self):
            for req in self.reqs:
                req.wait()
            for t, v in self.copies:
                v.copy_(t)
            self.reqs, self.copies = [], []

    def post(sends, recvs):
        if not sends and not recvs:
            return _Done()
        if staged_device is not None:
            outs = [(dst, v.cpu()) for dst, v in sends]
            ins = [(src, torch.empty(v.shape, dtype=v.dtype), v) for src, v in recvs]
            ops = [dist.P2POp(dist.isend, t, dst) for dst, t in outs] + [dist.P2POp(dist.irecv, t, src) for src, t, _ in ins]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            for _, t, v in ins:
                v.copy_(t)
            return _Done()
        # RCCL moves contiguous tensors only: the pieces are contiguous by construction (row slabs, whole buffers), but a view
        # that is not goes through a contiguous temporary rather than aborting the job
        outs = [(dst, v if v.is_contiguous() else v.contiguous()) for dst, v in sends]
        ins = [(src, v if v.is_contiguous() else torch.empty(v.shape, dtype=v.dtype, device=v.device), v) for src, v in recvs]
        ops = [dist.P2POp(dist.isend, t, dst) for dst, t in outs] + [dist.P2POp(dist.irecv, t, src) for src, t, _ in ins]
        return _Pending(dist.batch_isend_irecv(ops), [(t, v) for _, t, v in ins if t is not v])

    def exchange(sends, recvs):
        post(sends, recvs).wait()

    exchange.post = post
    return exchange


# ==================================================================================================
# Row slabs: the simplest distributed product, C_r = A_r * B with B replicated by ONE all-gather.
# Rank r holds rows [cut(r), cut(r+1)) of A, of B and (afterwards) of C -- the layout closes, products chain.
# The only communication is the all-gather of B's row slabs (an RCCL collective; on the xGMI mesh every link
# carries 1/W of B).  Each rank then multiplies its slab by the whole B with the full single-GPU engine; what the
# variant gives up is Strassen depth in the row direction (a slab of m/W rows has log2(W) levels fewer), which is
# why the Strassen-sharded variant overtakes it at 8 GPUs while the slabs win at 2 and 4.
# ==================================================================================================
def slab_rows(rows: int, world: int) -> int:
    """Rows of a full slab: ceil(rows / W)."""
    return -(-rows // world)


def slab_cuts(rows: int, world: int):
    """Row boundaries of the W slabs: slab r = rows [r k, (r+1) k) clipped to the matrix, k = ceil(rows / W).  When
    W does not divide `rows` the last slab is short (and with rows < W(W-1) trailing slabs may be empty): global row g
    always sits at position g of the gathered buffer, so a ragged size needs no second code path -- the ranks' buffers
    are padded to k rows and the gathered one to W k, the collective keeps equal pieces (all_gather_rows)."""
    k = slab_rows(rows, world)
    return [min(r * k, rows) for r in range(world + 1)]


def all_gather_rows(dist, full, mine, staged=False, async_op=False):
    """full (W * k rows) <- the ranks' `mine` (k rows each, the short last slab padded) in rank order:
    dist.all_gather_into_tensor, or its host-staged equivalent for backends that cannot move device tensors (gloo
    in the one-GPU tests).  async_op: return a handle at once -- under RCCL the collective runs on the communicator's
    stream and handle.wait() orders the current stream behind it (the host does not block), so a product that needs only
    the rank's OWN slab of B can be multiplied meanwhile (bench.py, slabs variant); staged / gloo: complete on return."""
    assert full.shape[0] == dist.get_world_size() * mine.shape[0], (tuple(full.shape), tuple(mine.shape))
    import torch
    if not staged:
        work = dist.all_gather_into_tensor(full.view(-1), mine.contiguous().view(-1), async_op=async_op)
        return work if async_op else None
    parts = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine.cpu() if mine.is_cuda else mine.contiguous())
    k = mine.shape[0]
    for r, t in enumerate(parts):
        full[r * k:(r + 1) * k].copy_(t)
    return _Done() if async_op else None


def slab_product_pieces(cuts, rank):
    """Inner-dimension pieces of C_r = A_r * B for the overlapped row-slab product, in the order they are multiplied: the
    rank's OWN row slab of B first (resident: needs nothing from the links), then what lies before and after it in the
    gathered B.  Returns [(k0, k1, own)] with empty pieces dropped."""
    k0, k1 = cuts[rank], cuts[rank + 1]
    out = [(k0, k1, True), (0, k0, False), (k1, cuts[-1], False)]
    return [p for p in out if p[1] > p[0]]


# what the halves of a product must keep for a SHARDED Strassen level to pay for its exchange: one 4096-row tile, 4096 columns and
# 8192 inner bits (the single-GPU engine itself splits down to 4096 inner bits since round 4; across links the bar stays higher)
ENGINE_MIN_HALF = (4096, 8192, 4096)


def default_variant(world: int, m: int = 0, l: int = 0, n: int = 0) -> str:
    """What `--variant auto` hands out to the ranks.

    * 2 ranks: row slabs (one all-gather of B; 2 ranks share a single link, which the Strassen-sharded exchange would saturate);
    * from 5 ranks on: the sub-products of the top Strassen level(s) -- but only for a product the single-GPU engine
      would itself split once more in all three dimensions.  A short inner dimension (BASELINE.json configs[4]:
      131072 x 8192 x 131072, l/2 < 8192) or a thin operand leaves nothing for a Strassen level to save: such shapes
      take row slabs of A and C with B replicated at every world size, no reduction (the reference's
      own row parallelism, m4ri/brilliantrussian.c:1121-1123);
    * 3 and 4 ranks: row slabs, except where two sharded levels are the 47 sub-products of the rank-47 scheme and those are at
      least 16384 on every side (65536^3 on 4 ranks: 12 sub-products per rank in batched products 6.58 ms, the row slab 8.00 ms:
      profiles/r06_rank_batch_timing.log) -- the rule of m4ri_amd_multi_default_variant (multi.hip), restated.
    The owner layout and the blocks variant scatter from rank 0 inside the timed region and are never selected here.
    """
    if world <= 2:
        return "slabs"
    if m and l and n and (m // 2 < ENGINE_MIN_HALF[0] or l // 2 < ENGINE_MIN_HALF[1] or n // 2 < ENGINE_MIN_HALF[2]):
        return "slabs"
    if world <= 4:
        if not (m and l and n) or min(m, l, n) // 4 < 16384:
            return "slabs"
        import m4ri_amd
        return "strassen" if m4ri_amd.shard_plan(world, m, l, n, 2).nprod != 49 else "slabs"
    return "strassen"
