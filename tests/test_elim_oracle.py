"""The oracle's table primitives pinned against the reference's mzd_make_table and mzd_process_rows{,2..6}
(m4ri/brilliantrussian.c:163-601), including tables whose source rows run past the end of the matrix."""
import numpy as np
import pytest

import elim_cases as ec
from m4ri_amd.mzd import Mzd


@pytest.mark.parametrize("nrows,ncols,r,c,k,nt", ec.CASES)
def test_tables_and_row_processing_match_reference(oracle, reference, nrows, ncols, r, c, k, nt):
    RL = ec.bind_reference(reference)
    M = Mzd.random(nrows, ncols, 7 * nrows + ncols + k)
    To, Lo = ec.tables_for(oracle.make_table, M, r, c, k, nt)
    Tr, Lr = ec.tables_for(lambda M_, r_, c_, k_, T, L: ec.call_make_table(RL, M_, r_, c_, k_, T, L), M, r, c, k, nt)
    for a, b, la, lb in zip(To, Tr, Lo, Lr):
        assert np.array_equal(a.rows(), b.rows()) and np.array_equal(la, lb)
    Mo, Mr = M.copy(), M.copy()
    oracle.process_rows(Mo, 0, nrows, c, k, To, Lo)
    ec.call_process_rows(RL, Mr, 0, nrows, c, k, Tr, Lr)
    assert np.array_equal(Mo.rows(), Mr.rows())
    Mo, Mr = M.copy(), M.copy()   # a row range, as the elimination loop uses it below the pivots
    oracle.process_rows(Mo, r + k if r + k < nrows else 0, nrows, c, k, To, Lo)
    ec.call_process_rows(RL, Mr, r + k if r + k < nrows else 0, nrows, c, k, Tr, Lr)
    assert np.array_equal(Mo.rows(), Mr.rows())


def test_make_table_keeps_stale_rows_like_the_reference(oracle, reference):
    """Source rows beyond the matrix: those steps are skipped and the table row keeps its previous content,
    which then seeds the following steps (brilliantrussian.c:181)."""
    RL = ec.bind_reference(reference)
    M = Mzd.random(20, 200, 3)
    for (r, k) in [(15, 8), (19, 4), (20, 3)]:
        To, Tr = Mzd.random(1 << k, 200, 9), Mzd.random(1 << k, 200, 9)   # dirty tables
        Lo, Lr = np.zeros(1 << k, dtype=np.int32), np.zeros(1 << k, dtype=np.int32)
        oracle.make_table(M, r, 70, k, To, Lo)
        ec.call_make_table(RL, M, r, 70, k, Tr, Lr)
        assert np.array_equal(To.rows(), Tr.rows()) and np.array_equal(Lo, Lr)
