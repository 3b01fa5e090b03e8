"""The host entry points' slab pipeline (mzd_api.hip: run_pipelined; m4ri_amd_set_host_pipeline): large products
from host memory are cut into a 2 x 2 grid of blocks of C (or four row slabs when n is small) so that PCIe copies run under the
products.  A block is an
ordinary product, so every bit must equal the one-shot schedule's and the oracle's -- ragged last slabs, windows
with non-zero excess (parent bits outside the window untouched), accumulate."""
import numpy as np
import pytest

import m4ri_amd
from m4ri_amd.mzd import Mzd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert m4ri_amd.lib().m4ri_amd_device_count() >= 1, "no HIP device visible: the gpu tests have nothing to run on"
    m4ri_amd.init(0)
    old = m4ri_amd.set_host_pipeline(1)   # every product with >= 16384 rows takes the pipeline
    yield
    m4ri_amd.set_host_pipeline(old)


@pytest.mark.parametrize("m,l,n", [(16384, 1000, 900), (16384 + 37, 2000, 1500), (20000, 257, 4100), (40000, 640, 640), (16385, 64, 1),
                                   (16384, 300, 16384), (20001, 700, 17000 + 13), (16500, 64, 30000),
                                   (33000, 300, 16384 + 1)])   # n >= 16384: the 2 x 2 grid; 33000 rows cut at 16384 (the coarse grid), 20001 at 12288
def test_pipelined_products_match_oracle(oracle, m, l, n):
    A, B = Mzd.random(m, l, 1), Mzd.random(l, n, 2)
    want = oracle.mul(None, A, B, 0)
    assert m4ri_amd.mzd_mul(None, A, B, 0).equal(want)
    C = Mzd.random(m, n, 3)
    assert m4ri_amd.mzd_mul(C, A, B, 0).equal(want)
    assert not (C.valid_words()[:, -1] & ~np.uint64(C.high_bitmask)).any()
    C0 = Mzd.random(m, n, 4)
    want2 = oracle.addmul(C0.copy(), A, B, 0)
    assert m4ri_amd.mzd_addmul(C0, A, B, 0).equal(want2)
    assert m4ri_amd._mzd_mul_even(Mzd.init(m, n), A, B, 256).equal(want)


def test_pipelined_windows_keep_their_parents(oracle):
    PA, PB, PC = Mzd.random(21000, 1200, 5), Mzd.random(1200, 1300, 6), Mzd.random(21000, 1300, 7)
    a, b, c = PA.window(100, 64, 100 + 18000, 64 + 1000), PB.window(7, 128, 7 + 1000, 128 + 1001), PC.window(900, 192, 900 + 18000, 192 + 1001)
    PCo = Mzd(21000, 1300, buf=PC.buf.copy())
    co = PCo.window(900, 192, 900 + 18000, 192 + 1001)
    oracle.mul(co, a.copy(), b.copy(), 0)
    m4ri_amd.mzd_mul(c, a, b, 0)
    assert np.array_equal(PC.buf, PCo.buf)
    oracle.addmul(co, a.copy(), b.copy(), 0)
    m4ri_amd.mzd_addmul(c, a, b, 0)
    assert np.array_equal(PC.buf, PCo.buf)
    # the 2 x 2 grid on windows: the column cut falls inside the windows of B and C
    QA, QB, QC = Mzd.random(17000, 600, 15), Mzd.random(600, 18000, 16), Mzd.random(17000, 18000, 17)
    a, b, c = QA.window(5, 0, 5 + 16500, 513), QB.window(3, 64, 3 + 513, 64 + 16999), QC.window(400, 128, 400 + 16500, 128 + 16999)
    QCo = Mzd(17000, 18000, buf=QC.buf.copy())
    co = QCo.window(400, 128, 400 + 16500, 128 + 16999)
    oracle.addmul(co, a.copy(), b.copy(), 0)
    m4ri_amd.mzd_addmul(c, a, b, 0)
    assert np.array_equal(QC.buf, QCo.buf)


def test_pipelined_equals_one_shot_at_32768():
    n = 32768
    A, B = Mzd.random(n, n, 8), Mzd.random(n, n, 9)
    piped = m4ri_amd.mzd_mul(None, A, B, 0)
    old = m4ri_amd.set_host_pipeline(0)
    try:
        whole = m4ri_amd.mzd_mul(None, A, B, 0)
    finally:
        m4ri_amd.set_host_pipeline(old)
    assert piped.equal(whole)
