"""The device twin of _mzd_add (reference m4ri/mzd.c:1471-1583), m4ri_amd_xor_dev, against the oracle's
gf2o_add (itself pinned to the reference's _mzd_add in test_oracle_vs_reference.py): last word written
under C's column mask with C's other bits kept (mzd.c:1489), in-place forms, operands of different
strides (windows of different parents), width 1..9 words (the reference special-cases 1..8)."""
import numpy as np
import pytest
import torch

import m4ri_amd
from m4ri_amd.mzd import Mzd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert m4ri_amd.lib().m4ri_amd_device_count() >= 1, "no HIP device visible: the gpu tests have nothing to run on"
    m4ri_amd.init(0)
    torch.cuda.set_device(0)


def _dev(parent: Mzd):
    """Device copy of a whole parent buffer (same layout) -> (tensor, rowstride)."""
    t = torch.from_numpy(parent.rows().copy().view(np.int64)).cuda()
    return t, parent.rowstride


def _ptr(t, stride, r0, c0):
    return t.data_ptr() + 8 * (r0 * stride + c0 // 64)


CASES = [  # (rows, ncols) of the operation, (r0, c0) of the window in each 96 x 704 / 80 x 1024 / 100 x 640 parent
    (1, 1), (3, 63), (5, 64), (7, 65), (16, 127), (16, 128), (33, 200), (64, 449), (20, 512), (11, 513), (9, 576), (40, 1),
]


@pytest.mark.parametrize("rows,ncols", CASES)
@pytest.mark.parametrize("alias", ["none", "c_is_a", "c_is_b", "a_is_b"])
def test_xor_dev_is_mzd_add(oracle, rows, ncols, alias):
    PA, PB, PC = Mzd.random(96, 704, 1), Mzd.random(80, 1024, 2), Mzd.random(100, 640, 3)  # pattern-filled parents
    offs = {"A": (7, 64), "B": (3, 128), "C": (11, 0)}
    wa = PA.window(offs["A"][0], offs["A"][1], offs["A"][0] + rows, offs["A"][1] + ncols)
    wb = PB.window(offs["B"][0], offs["B"][1], offs["B"][0] + rows, offs["B"][1] + ncols)
    wc = PC.window(offs["C"][0], offs["C"][1], offs["C"][0] + rows, offs["C"][1] + ncols)
    tA, sA = _dev(PA)
    tB, sB = _dev(PB)
    tC, sC = _dev(PC)
    pa, pb, pc = _ptr(tA, sA, *offs["A"]), _ptr(tB, sB, *offs["B"]), _ptr(tC, sC, *offs["C"])
    if alias == "none":
        oracle.add(wc, wa, wb)
        m4ri_amd.xor_dev(pc, sC, pa, sA, pb, sB, rows, ncols)
    elif alias == "c_is_a":
        oracle.add(wc, wc, wb)
        m4ri_amd.xor_dev(pc, sC, pc, sC, pb, sB, rows, ncols)
    elif alias == "c_is_b":
        oracle.add(wc, wa, wc)
        m4ri_amd.xor_dev(pc, sC, pa, sA, pc, sC, rows, ncols)
    else:
        oracle.add(wc, wa, wa)  # A + A = 0 on the valid bits, the rest of C's last word stays
        m4ri_amd.xor_dev(pc, sC, pa, sA, pa, sA, rows, ncols)
    torch.cuda.synchronize()
    got = tC.cpu().numpy().view(np.uint64)
    assert np.array_equal(got, PC.rows()), "the WHOLE parent of C must match: valid bits, excess bits, other columns"
    assert np.array_equal(tA.cpu().numpy().view(np.uint64), PA.rows()) and np.array_equal(tB.cpu().numpy().view(np.uint64), PB.rows())


def test_xor_dev_large_mixed_strides(oracle):
    """Quadrant-sized add as the Strassen levels use it (mzd.c:1471 called from strassen.c:111-150)."""
    P = Mzd.random(4100, 8200, 5)
    Q = Mzd.random(4100, 4100, 6)
    a, b = P.window(0, 0, 2050, 4037), P.window(2050, 4096, 4100, 8133)
    c = Q.window(1000, 0, 3050, 4037)
    tP, sP = _dev(P)
    tQ, sQ = _dev(Q)
    oracle.add(c, a, b)
    m4ri_amd.xor_dev(_ptr(tQ, sQ, 1000, 0), sQ, _ptr(tP, sP, 0, 0), sP, _ptr(tP, sP, 2050, 4096), sP, 2050, 4037)
    torch.cuda.synchronize()
    assert np.array_equal(tQ.cpu().numpy().view(np.uint64), Q.rows())
