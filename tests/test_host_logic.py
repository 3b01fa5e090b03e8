"""Host-side scheduling logic of the engine that needs no GPU: the Strassen-Winograd depth rule
(engine.hip plan_levels) against a restatement of the reference's recursion test (strassen.c:39,51,
71-80) and the engine's documented default; the multi-GPU per-rank blocks keep their depth."""
import numpy as np
import pytest

import m4ri_amd
from m4ri_amd import sharding


def ref_levels(m, l, n, cutoff):
    """strassen.c:39-51: _mzd_mul_even recurses on halves until closer(dim, cutoff) for any dimension;
    cutoff is normalised to a multiple of 64, >= 64 (strassen.c:351-354)."""
    cutoff = max(64, cutoff // 64 * 64)
    L = 0
    while not any(3 * d < 4 * cutoff for d in (m, l, n)) and L < 6:
        m, l, n, L = m // 2, l // 2, n // 2, L + 1
    return L


def cap(m, l, n, L):  # every level halves whole words of l and n, and rows of m
    while L > 0 and ((m >> L) == 0 or l // (64 << L) == 0 or n // (64 << L) == 0):
        L -= 1
    return L


@pytest.mark.parametrize("m,l,n,cutoff", [
    (65536, 65536, 65536, 4096), (16384, 16384, 16384, 4096), (4096, 4096, 4096, 4096), (131072, 8192, 131072, 4096),
    (2048, 2048, 4096, 1024), (4096, 3528, 4096, 1024), (1025, 1025, 1025, 256), (1000, 1000, 1000, 256),
    (1290, 1710, 2000, 256), (21, 171, 31, 63), (193, 65, 65, 64), (8192, 8192, 8192, 2048), (300, 300, 300, 64),
    (65536, 65536, 65536, 100), (1, 1, 1, 1024),
])
def test_caller_cutoff_follows_the_reference_rule(m, l, n, cutoff):
    assert m4ri_amd.plan_levels(m, l, n, cutoff) == cap(m, l, n, ref_levels(m, l, n, cutoff))


def test_reference_default_cutoff_depths_of_the_survey():
    # with the reference's default cutoff 4096: 4096 -> 0 levels, 16384 -> 2, 65536 -> 4
    assert [ref_levels(n, n, n, 4096) for n in (4096, 16384, 65536)] == [0, 2, 4]
    assert [m4ri_amd.plan_levels(n, n, n, 4096) for n in (4096, 16384, 65536)] == [0, 2, 4]


@pytest.mark.parametrize("shape,levels", [   # the engine's own depth: the minimum of its time model (engine.hip depth_model_seconds),
    # every row checked against the measured best depth (profiles/r04_depth_model_sweep.log, r04_depth_model_validation.log)
    ((65536, 65536, 65536), 4), ((32768, 32768, 32768), 3), ((16384, 16384, 16384), 2), ((8192, 8192, 8192), 0), ((4096, 4096, 4096), 0),
    ((131072, 131072, 131072), 5),
    ((131072, 8192, 131072), 2), ((131072, 16384, 131072), 4), ((262144, 8192, 32768), 2), ((131072, 8192, 8192), 2),   # short inner dimension, long rows: leaves of 2048 inner bits pay
    ((32768, 4096, 32768), 0), ((131072, 4096, 131072), 2),                                                          # ... leaves of 1024 do only on long rows (since the rank-47 scheme: 10.30 ms at L2 against 11.06 at L0, profiles/r05_depth_model_scheme47.log)
    ((16384, 65536, 65536), 2), ((16384, 16384, 65536), 2),                                                          # leaves keep a whole 4096-row tile
    ((16421, 16453, 16523), 0), ((50000, 12000, 90000), 0),                                                          # ragged: the strips cost more than a level saves
    ((24576, 24576, 24576), 1), ((49152, 49152, 49152), 2), ((57344, 57344, 57344), 3),                              # rows in whole tiles
    ((70000, 524288, 512), 0), ((65536, 65536, 1024), 0), ((1100, 1290, 1411), 0),
])
def test_engine_default_depth(shape, levels):
    assert m4ri_amd.plan_levels(*shape, 0) == levels


@pytest.mark.parametrize("shape,blocks", [   # rows that do not tile: the largest block of k * 4096 * 2^L rows at its own depth, the rest after it
    ((65664, 65664, 65664), [(65536, 4), (128, 0)]),            # 29.9 ms; the best single product takes 38.4 (profiles/r04_row_blocks_sweep.log)
    ((69632, 65536, 65536), [(65536, 4), (4096, 0)]),           # 29.9 against 37.7
    ((36864, 36864, 36864), [(32768, 2), (4096, 0)]),           # 6.23 against 6.90
    ((40960, 40960, 40960), [(32768, 3), (8192, 1)]),           # 8.51 against 9.02
    ((100003, 50021, 70017), [(98304, 3), (1699, 0)]),          # 41.3 against 47.8
    ((70000, 70000, 70000), [(65536, 3), (4464, 0)]),           # 41.0 against 45.5
    ((20480, 20480, 20480), [(16384, 2), (4096, 0)]),           # 1.22 against 1.45
    ((33000, 33000, 33000), [(32768, 3), (232, 0)]),            # 5.24 against 5.62
    ((73728, 16384, 65536), [(65536, 4), (8192, 1)]),           # 9.33 against 9.80 (round 4, L3); with the rank-47 scheme L4: 7.98 against 8.18 for the 65536-row block
    ((66000, 66000, 66000), [(65536, 4), (464, 0)]),            # 31.2 against 40.1
    ((20480, 65536, 65536), [(16384, 2), (4096, 0)]),           # 10.6 against 11.8
    ((69632, 8192, 131072), [(65536, 2), (4096, 0)]),           # 9.45 against 10.3
    ((65536, 65536, 65536), [(65536, 4)]), ((49152, 49152, 49152), [(49152, 2)]), ((45000, 45000, 45000), [(45000, 2)]),   # one product stays one product
    ((50000, 12000, 90000), [(50000, 0)]), ((34000, 20000, 20000), [(34000, 0)]),   # blocks measured 10 and 16 % slower: a separate pack pass, strips twice
])
def test_engine_row_blocks(shape, blocks):
    got = m4ri_amd.plan_row_blocks(*shape)
    assert got == blocks and sum(r for r, _ in got) == shape[0]
    assert m4ri_amd.plan_levels(*shape, 0) == blocks[0][1]


def test_engine_default_depth_properties():
    """What must hold for ANY shape, whatever the model's constants: leaves keep a whole 4096-row tile and 1024 inner bits and
    columns (or the product stays unsplit), and the depth of a power-of-two cube from 16384 up leaves 4096^3 leaves (the
    reference's own default depth)."""
    rng = np.random.default_rng(5)
    for _ in range(400):
        m, l, n = (int(x) for x in np.exp(rng.uniform(np.log(64), np.log(300000), 3)))
        L = m4ri_amd.plan_levels(m, l, n, 0)
        assert 0 <= L <= 6
        if L:
            assert (m >> L) >= 4096 and (l >> L) >= 1024 and (n >> L) >= 1024, (m, l, n, L)
        blocks = m4ri_amd.plan_row_blocks(m, l, n)
        assert sum(r for r, _ in blocks) == m and all(r > 0 for r, _ in blocks) and blocks[0][1] == L
        for r, lv in blocks[:-1]:      # every block but the last is whole tiles of rows at its depth
            assert lv >= 1 and r % (4096 << 1) == 0, (m, l, n, blocks)
        for r, lv in blocks:
            assert lv == 0 or (r >> lv) >= 4096
    for k in range(14, 19):
        assert m4ri_amd.plan_levels(1 << k, 1 << k, 1 << k, 0) == k - 12


def test_per_rank_blocks_of_the_default_grids_keep_their_depth():
    n = 65536
    want = {1: 4, 2: 3, 4: 3, 8: 2}
    for world, L in want.items():
        p = sharding.make_plan(world, 0, n, n, n)
        (r0, r1), (c0, c1), (k0, k1) = p.row_range(), p.col_range(), p.inner_range()
        assert m4ri_amd.plan_levels(r1 - r0, k1 - k0, c1 - c0, 0) == L


# ---- the multi-GPU schedules behind the C boundary: their host arithmetic (multi.hip, no GPU needed) -----------------------
def test_schedule_choice_in_c_matches_the_python_rule():
    """m4ri_amd_multi_default_variant is sharding.default_variant moved behind the C boundary: same answer for every world size
    and shape class (BASELINE.json configs[3] -> Strassen sub-products at 8 ranks, configs[4] -> row slabs at every world size)."""
    names = {m4ri_amd.VARIANT_SLABS: "slabs", m4ri_amd.VARIANT_STRASSEN: "strassen"}
    for world in (1, 2, 3, 4, 5, 7, 8, 16):
        for shape in [(65536, 65536, 65536), (131072, 8192, 131072), (16384, 16384, 16384), (8192, 65536, 65536), (65536, 16384, 8192),
                      (100003, 50021, 70017), (8190, 16384, 8192)]:
            assert names[m4ri_amd.multi_default_variant(world, *shape)] == sharding.default_variant(world, *shape), (world, shape)
    assert m4ri_amd.multi_layout_for(0, 8, 65536, 65536, 65536) == m4ri_amd.LAYOUT_CYCLIC2   # 47 sub-products (the scheme once) over 8 ranks: 6 rounds
    assert m4ri_amd.multi_layout_for(0, 7, 65536, 65536, 65536) == m4ri_amd.LAYOUT_CYCLIC1   # 7 sub-products, one per rank
    assert m4ri_amd.multi_layout_for(0, 8, 131072, 8192, 131072) == m4ri_amd.LAYOUT_ROWS
    assert m4ri_amd.multi_layout_for(m4ri_amd.VARIANT_STRASSEN, 4, 65536, 65536, 65536) == m4ri_amd.LAYOUT_CYCLIC2   # 47 products over 4 ranks


@pytest.mark.parametrize("layout", [m4ri_amd.LAYOUT_ROWS, m4ri_amd.LAYOUT_CYCLIC1, m4ri_amd.LAYOUT_CYCLIC2])
@pytest.mark.parametrize("world", [1, 2, 3, 7, 8, 64])
def test_layout_runs_partition_the_rows(layout, world):
    """Every valid row of a distributed matrix lives on exactly one rank (ROWS, CYCLIC1, CYCLIC2), inside that rank's local buffer,
    and the local buffers of the CYCLIC layouts are whole slabs of every row block."""
    for rows in (1, 63, 256, 1000, 4097, 65536, 100003):
        seen = [0] * rows if rows <= 5000 else None
        total = 0
        for r in range(world):
            lrows = m4ri_amd.lib().m4ri_amd_layout_local_rows(layout, world, r, rows)
            for g0, n, l0 in m4ri_amd.layout_runs(layout, world, r, rows):
                assert n > 0 and 0 <= g0 and g0 + n <= rows and 0 <= l0 and l0 + n <= lrows
                total += n
                if seen is not None:
                    for g in range(g0, g0 + n):
                        seen[g] += 1
        assert total == rows and (seen is None or set(seen) == {1}), (layout, world, rows)
    assert m4ri_amd.layout_runs(m4ri_amd.LAYOUT_REPLICATED, world, world - 1, 77) == [(0, 77, 0)]


def test_pair_table_of_the_multi_device_path():
    """Which ordered rank pairs copy through the host (m4ri_amd_multi_pair_table, the rule ensure_ranks applies before the first
    product): pairs whose DEVICES have no peer access, plus the test hook's rank pairs; ranks sharing a device copy directly."""
    full = [[1] * 4 for _ in range(4)]
    assert m4ri_amd.multi_pair_table([0, 1, 2, 3], full) == [[0] * 4 for _ in range(4)]
    # device 1 cannot map device 3 (one direction only): exactly the pair rank(dev 1) <- rank(dev 3)
    can = [row[:] for row in full]
    can[1][3] = 0
    t = m4ri_amd.multi_pair_table([0, 1, 2, 3], can)
    assert t[1][3] == 1 and sum(map(sum, t)) == 1
    # eight ranks on four devices, two per device: the device pair (1, 3) is four rank pairs; same-device ranks stay direct
    t = m4ri_amd.multi_pair_table([0, 0, 1, 1, 2, 2, 3, 3], can)
    assert sorted((i, j) for i in range(8) for j in range(8) if t[i][j]) == [(2, 6), (2, 7), (3, 6), (3, 7)]
    # the hook: "all" = every pair but the diagonal, a list = those rank pairs in both directions; junk and out-of-range entries are ignored
    assert sum(map(sum, m4ri_amd.multi_pair_table([0] * 4, [[1]], "all"))) == 12
    t = m4ri_amd.multi_pair_table([0] * 4, [[1]], "0-2,1-3")
    assert sorted((i, j) for i in range(4) for j in range(4) if t[i][j]) == [(0, 2), (1, 3), (2, 0), (3, 1)]
    assert sum(map(sum, m4ri_amd.multi_pair_table([0] * 4, [[1]], "0-9,2-2,x"))) == 0
    # no access anywhere: every pair of different devices
    t = m4ri_amd.multi_pair_table([0, 1, 2], [[1, 0, 0], [0, 1, 0], [0, 0, 1]])
    assert t == [[0, 1, 1], [1, 0, 1], [1, 1, 0]]


def test_the_4x4x4_scheme_is_a_scheme():
    """m4ri_amd/csrc/scheme444.h (generated by tools/make_scheme_header.py from what tools/flipgraph_444.c found): R rank-one tensors (u, v, w)
    over GF(2) that sum to the tensor of the 4 x 4 x 4 matrix product -- checked here against the definition, entry by entry, because every
    product of four fused Strassen levels goes through this table (scheme_passes.hip).  R <= 49 (Strassen applied twice)."""
    import os
    import re
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "m4ri_amd", "csrc", "scheme444.h")).read()
    R = int(re.search(r"#define SCHEME444_R (\d+)", text).group(1))
    tabs = {name: [int(x, 16) for x in re.findall(r"0x([0-9a-f]{4})", re.search(rf"SCHEME444_{name}\[SCHEME444_R\] = \{{([^}}]*)\}}", text).group(1))] for name in "UVW"}
    assert all(len(t) == R for t in tabs.values()) and 40 <= R <= 49 and all(x > 0 for t in tabs.values() for x in t)
    U, V, W = (np.array(tabs[k], dtype=np.uint32) for k in "UVW")
    bit = lambda a, k: ((a >> np.uint32(k)) & np.uint32(1)).astype(np.uint8)   # noqa: E731
    for i in range(4):
        for j in range(4):
            for j2 in range(4):
                for k in range(4):
                    for i2 in range(4):
                        for k2 in range(4):
                            got = int((bit(U, 4 * i + j) & bit(V, 4 * j2 + k) & bit(W, 4 * i2 + k2)).sum() & 1)
                            assert got == int(i == i2 and j == j2 and k == k2), (i, j, j2, k, i2, k2)


def test_the_winograd_level_over_the_scheme_is_a_scheme():
    """The outer tables of the three-level scheme passes (scheme_passes.hip: make_tables, WG) are Winograd's level in the engine's order; as
    rank-one tensors over the 2 x 2 quadrants they must sum to the tensor of the 2 x 2 block product."""
    import os
    import re
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "m4ri_amd", "csrc", "scheme_passes.hip")).read()
    rows = re.search(r"constexpr uint16_t WG\[3\]\[7\] = \{\{([^}]*)\}, \{([^}]*)\}, \{([^}]*)\}\};", text).groups()
    U, V, W = ([int(x, 16) for x in re.findall(r"0x([0-9A-Fa-f]+)", row)] for row in rows)
    assert len(U) == len(V) == len(W) == 7
    for i in range(2):
        for j in range(2):
            for j2 in range(2):
                for k in range(2):
                    for i2 in range(2):
                        for k2 in range(2):
                            got = sum((u >> (2 * i + j)) & (v >> (2 * j2 + k)) & (w >> (2 * i2 + k2)) & 1 for u, v, w in zip(U, V, W)) & 1
                            assert got == int(i == i2 and j == j2 and k == k2)


def test_a_rank_multiplies_its_sub_products_in_groups():
    """m4ri_amd_shard_group / m4ri_amd_model_seconds_batch (pure arithmetic): the sub-products a rank of the sharded Strassen schedule owns go
    into batched products (m4ri_amd_mul_batch_dev) -- the smallest group within 5 % of the time model's best; measured on one MI355X for the
    8-rank split of 65536^3: 6 x 16384^3 one at a time 3.54 ms, 2 + 2 + 2 3.31, all six 3.33 (profiles/r06_rank_batch_timing.log)."""
    p = m4ri_amd.shard_plan(8, 65536, 65536, 65536)
    assert (p.levels, p.nprod) == (2, 47) and m4ri_amd.shard_group(p) == 2
    assert m4ri_amd.shard_group(p, 4096) == 1                                          # a caller's cutoff: one at a time
    assert m4ri_amd.shard_group(m4ri_amd.shard_plan(8, 65536, 65536, 65536, 1)) == 1   # one sub-product per rank
    assert m4ri_amd.shard_group(m4ri_amd.shard_plan(8, 131072, 131072, 131072)) == 1   # sub-products of 32768^3 fill the chip alone
    one, two, six = (m4ri_amd.model_seconds_batch(16384, 16384, 16384, -1, b) for b in (1, 2, 6))
    assert two < 1.9 * one and six < 5.5 * one and six > 4.0 * one                      # half-filled last rounds of tiles become full ones
    for L in (0, 1, 2, 3):
        assert m4ri_amd.model_seconds_batch(16384, 16384, 16384, L, 1) == m4ri_amd.model_seconds(16384, 16384, 16384, L)


def test_which_direct_products_take_the_small_leaf_and_how_they_are_split():
    """engine.hip: small_leaf_wanted + m4rm_small.hip: gf2_m4rm_small_ksplit, through m4ri_amd_plan_small_leaf (pure arithmetic).  The rule:
    direct products of up to 2^34 bit operations, batch included; the inner splits minimise rounds x steps x 1.9 us + 38.5 ps per word of C
    and split -- pinned here against the picks that were measured on one MI355X (profiles/r06_small_leaf_vs_generation4.log)."""
    import os
    if os.environ.get("M4RI_AMD_SMALL_LEAF") or os.environ.get("M4RI_AMD_SMALL_LEAF_WORK"):
        pytest.skip("the small-leaf rule is overridden by the environment")
    plan = m4ri_amd.lib().m4ri_amd_plan_small_leaf
    for (m, l, n, batch, want) in [(512, 512, 512, 1, 8), (2048, 2048, 2048, 1, 4), (2560, 2560, 2560, 1, 4), (64, 1 << 20, 64, 1, 256),
                                   (4096, 256, 4096, 1, 1), (1, 1, 1, 1, 1), (64, 64, 64, 1, 1)]:
        assert plan(m, l, n, batch, 256) == want, (m, l, n, batch, plan(m, l, n, batch, 256))
    assert 4 <= plan(1536, 1536, 1536, 1, 256) <= 8 and 12 <= plan(4096, 4096, 256, 1, 256) <= 16 and 12 <= plan(8192, 8192, 200, 1, 256) <= 16   # flat optima
    for (m, l, n, batch) in [(4096, 4096, 4096, 1), (3072, 3072, 3072, 1), (2048, 2048, 2048, 3), (464, 16384, 16421, 1), (0, 5, 5, 1), (5, 0, 5, 1)]:
        assert plan(m, l, n, batch, 256) == 0, (m, l, n, batch)     # above 2^34 bit operations (batch included), or nothing to multiply
    assert plan(1024, 1024, 1024, 4, 256) >= 1 and plan(1024, 1024, 1024, 32, 256) == 0
    # fewer compute units, fewer splits; a split count is always one the launcher keeps (whole steps per split)
    assert plan(512, 512, 512, 1, 8) <= plan(512, 512, 512, 1, 256)
    for l in range(64, 64 * 40, 64):
        k = plan(256, l, 256, 1, 256)
        steps = -(-(l // 64) // k)
        assert k >= 1 and -(-(l // 64) // steps) == k, (l, k)
