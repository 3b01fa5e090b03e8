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
