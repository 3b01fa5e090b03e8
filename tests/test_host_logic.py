"""Host-side scheduling logic of the engine that needs no GPU: the Strassen-Winograd depth rule
(engine.hip plan_levels) against a restatement of the reference's recursion test (strassen.c:39,51,
71-80) and the engine's documented default; the multi-GPU per-rank blocks keep their depth."""
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


@pytest.mark.parametrize("shape,levels", [
    ((65536, 65536, 65536), 3), ((32768, 32768, 32768), 2), ((16384, 16384, 16384), 1), ((8192, 8192, 8192), 0),
    ((131072, 131072, 131072), 4), ((131072, 8192, 131072), 0), ((16421, 16453, 16523), 1),
])
def test_engine_default_depth(shape, levels):
    assert m4ri_amd.plan_levels(*shape, 0) == levels


def test_per_rank_blocks_of_the_default_grids_keep_their_depth():
    n = 65536
    want = {1: 3, 2: 3, 4: 3, 8: 2}
    for world, L in want.items():
        p = sharding.make_plan(world, 0, n, n, n)
        (r0, r1), (c0, c1), (k0, k1) = p.row_range(), p.col_range(), p.inner_range()
        assert m4ri_amd.plan_levels(r1 - r0, k1 - k0, c1 - c0, 0) == L
