"""mzd_make_table and mzd_process_rows{,2..6} on the GPU (include/m4ri_amd.h; reference
m4ri/brilliantrussian.c:163-601) against the oracle's restatement (pinned to the reference in
tests/test_elim_oracle.py): table contents, L arrays and the processed matrix, bit for bit, host matrices and
matrices pinned on the device."""
import numpy as np
import pytest

import elim_cases as ec
import m4ri_amd
from m4ri_amd.mzd import Mzd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert m4ri_amd.lib().m4ri_amd_device_count() >= 1, "no HIP device visible: the gpu tests have nothing to run on"
    m4ri_amd.init(0)


def _gpu_make(M, r, c, k, T, L):
    ec.call_make_table(m4ri_amd.lib(), M, r, c, k, T, L)


@pytest.mark.parametrize("nrows,ncols,r,c,k,nt", ec.CASES + [(20000, 8192, 100, 4096, 48, 6), (5000, 20000, 64, 0, 40, 5)])
def test_tables_and_row_processing_match_oracle(oracle, nrows, ncols, r, c, k, nt):
    M = Mzd.random(nrows, ncols, 7 * nrows + ncols + k)
    To, Lo = ec.tables_for(oracle.make_table, M, r, c, k, nt)
    Tg, Lg = ec.tables_for(_gpu_make, M, r, c, k, nt)
    for a, b, la, lb in zip(To, Tg, Lo, Lg):
        assert np.array_equal(a.rows(), b.rows()) and np.array_equal(la, lb)
    for (s0, s1) in [(0, nrows), (r + k if r + k < nrows else 0, nrows), (1, max(1, nrows // 2))]:
        Mo, Mg = M.copy(), M.copy()
        oracle.process_rows(Mo, s0, s1, c, k, To, Lo)
        ec.call_process_rows(m4ri_amd.lib(), Mg, s0, s1, c, k, Tg, Lg)
        assert np.array_equal(Mo.rows(), Mg.rows()), (s0, s1)


def test_stale_table_rows(oracle):
    M = Mzd.random(20, 200, 3)
    for (r, k) in [(15, 8), (19, 4), (20, 3)]:
        To, Tg = Mzd.random(1 << k, 200, 9), Mzd.random(1 << k, 200, 9)
        Lo, Lg = np.zeros(1 << k, dtype=np.int32), np.zeros(1 << k, dtype=np.int32)
        oracle.make_table(M, r, 70, k, To, Lo)
        _gpu_make(M, r, 70, k, Tg, Lg)
        assert np.array_equal(To.rows(), Tg.rows()) and np.array_equal(Lo, Lg)


def test_row_processing_on_a_pinned_matrix(oracle):
    """The elimination loop's pattern with the matrix resident on the device: several strips in a row."""
    M = Mzd.random(3000, 2048, 5)
    Mo = M.copy()
    m4ri_amd.pin(M)
    for (r, c, k, nt) in [(0, 0, 48, 6), (48, 48, 40, 5), (88, 88, 24, 3)]:
        T, L = ec.tables_for(oracle.make_table, Mo, r, c, k, nt)   # same tables on both sides (from the oracle's state)
        oracle.process_rows(Mo, r + k, 3000, c, k, T, L)
        ec.call_process_rows(m4ri_amd.lib(), M, r + k, 3000, c, k, T, L)
    m4ri_amd.unpin(M)
    assert np.array_equal(M.rows(), Mo.rows())


def test_edge_ranges_and_small_k(oracle):
    """Empty row ranges, the last column block, k = 1 and k = 2 tables."""
    M = Mzd.random(100, 200, 1)
    T, L = ec.tables_for(oracle.make_table, M, 3, 5, 2, 1)
    Mg = M.copy()
    ec.call_process_rows(m4ri_amd.lib(), Mg, 50, 50, 5, 2, T, L)   # startrow == stoprow: nothing happens
    assert np.array_equal(Mg.rows(), M.rows())
    for (r, c, k, nt) in [(0, 0, 1, 1), (10, 191, 8, 1), (10, 192, 8, 2), (0, 130, 6, 3), (90, 100, 2, 2)]:
        To, Lo = ec.tables_for(oracle.make_table, M, r, c, k, nt)
        Tg, Lg = ec.tables_for(_gpu_make, M, r, c, k, nt)
        for a, b, la, lb in zip(To, Tg, Lo, Lg):
            assert np.array_equal(a.rows(), b.rows()) and np.array_equal(la, lb)
        Mo, Mg = M.copy(), M.copy()
        oracle.process_rows(Mo, 0, 100, c, k, To, Lo)
        ec.call_process_rows(m4ri_amd.lib(), Mg, 0, 100, c, k, Tg, Lg)
        assert np.array_equal(Mo.rows(), Mg.rows()), (r, c, k, nt)


@pytest.mark.parametrize("pin_m", [False, True])
def test_tables_pinned_on_the_device(oracle, pin_m):
    """mzd_make_table into a PINNED T (and from a pinned M): the table is built inside T's device copy -- which is what
    mzd_process_rows then reads -- and T's host copy is stale until sync.  (Round 2 seeded the table from, and wrote it back
    to, the host copy only: the row processing that followed read a stale device copy.)"""
    nrows, ncols, r, c, k, nt = 600, 1500, 40, 130, 24, 3
    M = Mzd.random(nrows, ncols, 77)
    Mo = M.copy()
    kb = ec.split_k(k, nt)
    Tg = [Mzd.random(1 << b, ncols, 90 + i) for i, b in enumerate(kb)]     # dirty tables: stale rows / seeds must come from HERE
    To = [t.copy() for t in Tg]
    for t, o in zip(Tg, To):
        o.rows()[:, :] = t.rows()
        m4ri_amd.pin(t)
    if pin_m:
        m4ri_amd.pin(M)
    Lg = [np.zeros(1 << b, dtype=np.int32) for b in kb]
    Lo = [np.zeros(1 << b, dtype=np.int32) for b in kb]
    off = 0
    for t in range(nt):
        oracle.make_table(Mo, r + off, c + off, kb[t], To[t], Lo[t])
        _gpu_make(M, r + off, c + off, kb[t], Tg[t], Lg[t])
        assert m4ri_amd.is_pinned(Tg[t]) == 2, "a pinned table is built on the device: its host copy is stale"
        off += kb[t]
    oracle.process_rows(Mo, r + k, nrows, c, k, To, Lo)
    ec.call_process_rows(m4ri_amd.lib(), M, r + k, nrows, c, k, Tg, Lg)
    if pin_m:
        m4ri_amd.unpin(M)
    for t in Tg:
        m4ri_amd.unpin(t)
    for a, b, la, lb in zip(To, Tg, Lo, Lg):
        assert np.array_equal(a.rows(), b.rows()) and np.array_equal(la, lb)
    assert np.array_equal(M.rows(), Mo.rows())
