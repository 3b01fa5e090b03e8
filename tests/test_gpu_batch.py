"""m4ri_amd_m4rm_batch_dev (include/m4ri_amd.h): `batch` products of one shape in one launch, each against the oracle's
gf2o_mul (reference strassen.c:345-365 / brilliantrussian.c:1032-1190 semantics: only C's bits are observable)."""
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


@pytest.mark.parametrize("m,l,n,batch", [(1, 1, 1, 3), (64, 64, 64, 7), (300, 200, 130, 5), (512, 512, 512, 16), (1024, 1024, 1024, 4), (200, 4100, 70, 3),
                                         (4096, 512, 640, 2), (191, 65, 129, 9)])
@pytest.mark.parametrize("add", [0, 1])
def test_batched_products_match_oracle(oracle, m, l, n, batch, add):
    wl, wn = (l + 63) // 64, (n + 63) // 64
    sa, sb, sc = wl + 2, wn + (wn & 1), wn + 4          # padded strides, B's even
    abs_, bbs, cbs = m * sa + 6, l * sb, m * sc + 2      # batch strides with gaps
    A = [Mzd.random(m, l, 10 + b) for b in range(batch)]
    B = [Mzd.random(l, n, 50 + b) for b in range(batch)]
    C = [Mzd.random(m, n, 90 + b) for b in range(batch)]
    hA = np.zeros(batch * abs_, dtype=np.uint64); hB = np.zeros(batch * bbs, dtype=np.uint64); hC = np.zeros(batch * cbs, dtype=np.uint64)
    for b in range(batch):
        hA[b * abs_: b * abs_ + m * sa].reshape(m, sa)[:, :wl] = A[b].valid_words()
        hB[b * bbs: b * bbs + l * sb].reshape(l, sb)[:, :wn] = B[b].valid_words()
        hC[b * cbs: b * cbs + m * sc].reshape(m, sc)[:, :wn] = C[b].valid_words()
    tA, tB, tC = (torch.from_numpy(x.view(np.int64)).cuda() for x in (hA, hB, hC))
    rc = m4ri_amd.lib().m4ri_amd_m4rm_batch_dev(tC.data_ptr(), sc, cbs, tA.data_ptr(), sa, abs_, tB.data_ptr(), sb, bbs, m, l, n, batch, add, None)
    assert rc == 0
    torch.cuda.synchronize()
    got = tC.cpu().numpy().view(np.uint64)
    for b in range(batch):
        want = C[b].copy() if add else Mzd(m, n)
        (oracle.addmul if add else oracle.mul)(want, A[b], B[b], 0)
        assert np.array_equal(got[b * cbs: b * cbs + m * sc].reshape(m, sc)[:, :wn], want.valid_words()), (b, add)


# ---- m4ri_amd_mul_batch_dev: the batch WITH Strassen-Winograd levels (the several sub-products a rank of a sharded level owns) -------
def _filled(rows, ncols, stride, batch, bs, seed0):
    """`batch` matrices on the device, member b filled from splitmix64 seed seed0 + b, stride / batch stride in words (gaps stay zero)."""
    t = torch.zeros(batch * bs, dtype=torch.int64, device="cuda")
    for b in range(batch):
        m4ri_amd.fill_dev(t.data_ptr() + 8 * b * bs, stride, rows, ncols, seed0 + b, 0)
    return t


@pytest.mark.parametrize("m,l,n,batch,cutoff,add", [
    (1024, 1024, 1024, 4, 0, 0),          # no levels: one leaf launch for the batch
    (1024, 1024, 1024, 3, 256, 0),        # two levels forced by the caller's cutoff (the Winograd passes: leaves too small for the scheme)
    (2048, 1024, 1536, 2, 256, 1),        # accumulate, rectangular
    (2048, 2048, 2048, 3, 256, 0),        # three levels
    (1000, 1100, 1200, 3, 256, 0),        # ragged: one product at a time behind the same entry point
])
def test_batched_strassen_products_match_oracle(oracle, m, l, n, batch, cutoff, add):
    wl, wn = (l + 63) // 64, (n + 63) // 64
    sa, sb, sc = wl + 2, wn + (wn & 1), wn + 4
    abs_, bbs, cbs = m * sa + 6, l * sb, m * sc + 2
    A = [Mzd.random(m, l, 110 + b) for b in range(batch)]
    B = [Mzd.random(l, n, 150 + b) for b in range(batch)]
    C = [Mzd.random(m, n, 190 + b) for b in range(batch)]
    hA = np.zeros(batch * abs_, dtype=np.uint64); hB = np.zeros(batch * bbs, dtype=np.uint64); hC = np.zeros(batch * cbs, dtype=np.uint64)
    for b in range(batch):
        hA[b * abs_: b * abs_ + m * sa].reshape(m, sa)[:, :wl] = A[b].valid_words()
        hB[b * bbs: b * bbs + l * sb].reshape(l, sb)[:, :wn] = B[b].valid_words()
        hC[b * cbs: b * cbs + m * sc].reshape(m, sc)[:, :wn] = C[b].valid_words()
    tA, tB, tC = (torch.from_numpy(x.view(np.int64)).cuda() for x in (hA, hB, hC))
    m4ri_amd.mul_batch_dev(tC.data_ptr(), sc, cbs, tA.data_ptr(), sa, abs_, tB.data_ptr(), sb, bbs, m, l, n, batch, bool(add), cutoff, 0)
    torch.cuda.synchronize()
    got = tC.cpu().numpy().view(np.uint64)
    for b in range(batch):
        want = C[b].copy() if add else Mzd(m, n)
        (oracle.addmul if add else oracle.mul)(want, A[b], B[b], 0)
        assert np.array_equal(got[b * cbs: b * cbs + m * sc].reshape(m, sc)[:, :wn], want.valid_words()), (b, add)


@pytest.mark.parametrize("m,l,n,batch,add", [
    (8192, 8192, 8192, 3, 0),             # one level
    (16384, 16384, 16384, 2, 0),          # two levels = the rank-R scheme once; what a rank of the 8-GPU schedule multiplies
    (16384, 16384, 16384, 6, 0),          # ... all six of its sub-products in one batch
    (16384, 8192, 16384, 3, 1),           # rectangular, accumulate
    (32768, 32768, 32768, 2, 0),          # three levels (a Winograd level over the scheme)
    (8256, 8192, 8192, 2, 0),             # rows worth cutting into blocks: one at a time
])
def test_batched_strassen_products_equal_the_single_products(m, l, n, batch, add):
    """At sizes the oracle does not finish in seconds: the batch against the same products one call each (m4ri_amd_mul_dev, itself checked
    against the reference's fingerprints at these sizes in test_gpu_parity.py) -- bit-identical, and the stats say the batch was one launch."""
    wl, wn = l // 64, n // 64
    abs_, bbs, cbs = m * wl + 32, l * wn + 64, m * wn + 32
    tA, tB = _filled(m, l, wl, batch, abs_, 1000), _filled(l, n, wn, batch, bbs, 2000)
    tC = _filled(m, n, wn, batch, cbs, 3000)
    tR = tC.clone()
    m4ri_amd.mul_batch_dev(tC.data_ptr(), wn, cbs, tA.data_ptr(), wl, abs_, tB.data_ptr(), wn, bbs, m, l, n, batch, bool(add), 0, 0)
    st = m4ri_amd.get_stats()
    for b in range(batch):
        m4ri_amd.mul_dev(tR.data_ptr() + 8 * b * cbs, wn, tA.data_ptr() + 8 * b * abs_, wl, tB.data_ptr() + 8 * b * bbs, wn, m, l, n, bool(add), 0, 0)
    torch.cuda.synchronize()
    assert torch.equal(tC, tR)
    if m % 4096 == 0:
        assert int(st.leaf_launches) == 1 and int(st.leaf_products) % batch == 0 and int(st.leaf_products) >= batch, (st.leaf_launches, st.leaf_products)
