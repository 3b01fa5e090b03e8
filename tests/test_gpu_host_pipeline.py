"""The host entry points' slab pipeline (mzd_api.hip: run_pipelined; m4ri_amd_set_host_pipeline): large products
from host memory are cut into four row slabs of A and C so that PCIe copies run under the products.  A slab is an
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


@pytest.mark.parametrize("m,l,n", [(16384, 1000, 900), (16384 + 37, 2000, 1500), (20000, 257, 4100), (40000, 640, 640), (16385, 64, 1)])
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
