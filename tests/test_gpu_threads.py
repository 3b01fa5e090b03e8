"""The threading contract of the drop-in boundary (SURVEY.md 8(b) "Threading"): the library is callable from several host threads
at once on disjoint matrices -- what the reference allows in its thread-safe / OpenMP builds (configure.ac:111-121, m4ri/mmc.c:49-52)
-- and serialises the device work internally.  ctypes releases the GIL around every foreign call, so the Python threads below are
inside libm4ri_amd.so at the same time.  Everything bit for bit against the oracle.  (The same pattern runs under ThreadSanitizer in
tools/tsan_threads.cpp; its report is kept in profiles/.)"""
import threading

import numpy as np
import pytest
import torch  # noqa: F401 -- before libm4ri_amd.so: the process gets ONE HIP runtime, the one torch ships

import m4ri_amd
from m4ri_amd.mzd import Mzd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert m4ri_amd.lib().m4ri_amd_device_count() >= 1, "no HIP device visible: the gpu tests have nothing to run on"
    m4ri_amd.init(0)
    yield
    m4ri_amd.set_devices([])
    m4ri_amd.set_multi_threshold(16384)


def _run_threads(workers):
    errors = []

    def guard(fn, i):
        try:
            m4ri_amd.init(0)          # the HIP device is a per-thread binding
            fn(i)
        except BaseException as e:    # noqa: BLE001 -- reported in the main thread
            errors.append((i, repr(e)))
    th = [threading.Thread(target=guard, args=(fn, i)) for i, fn in enumerate(workers)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_four_threads_mul_addmul_m4rm_on_disjoint_matrices(oracle):
    shapes = [(1100, 1290, 1411), (2048, 2048, 4096), (513, 700, 65), (4096, 3528, 4096)]
    cases = []
    for i, (m, l, n) in enumerate(shapes):
        A, B, C0 = Mzd.random(m, l, 100 + i), Mzd.random(l, n, 200 + i), Mzd.random(m, n, 300 + i)
        cases.append((A, B, C0, oracle.mul(None, A, B, 0), oracle.addmul(C0.copy(), A, B, 0)))

    def work(i):
        A, B, C0, want_mul, want_add = cases[i]
        for rep in range(6):
            assert m4ri_amd.mzd_mul(None, A, B, 0 if rep % 2 else 256).equal(want_mul), ("mul", i, rep)
            assert m4ri_amd.mzd_addmul(C0.copy(), A, B, 0).equal(want_add), ("addmul", i, rep)
            assert m4ri_amd.mzd_mul_m4rm(None, A, B, 0).equal(want_mul), ("m4rm", i, rep)
            assert m4ri_amd._mzd_addmul(C0.copy(), A, B, 512).equal(want_add), ("_addmul", i, rep)
    _run_threads([work] * 4)


def test_threads_with_pinned_chains_and_windows(oracle):
    """Every thread pins its own matrices and chains products on the device (C += A*B twice gives C back; a window product
    lands inside the pinned parent) while the other threads do the same: the residency table is shared state."""
    n = 1536
    data = []
    for i in range(4):
        A, B, C = Mzd.random(n, n, 400 + i), Mzd.random(n, n, 500 + i), Mzd.random(n, n, 600 + i)
        want = oracle.addmul(C.copy(), A, B, 0)
        wa, wb = A.window(0, 0, 512, 1024), B.window(0, 0, 1024, 768)
        data.append((A, B, C, want, oracle.mul(None, wa.copy(), wb.copy(), 0)))

    def work(i):
        A, B, C, want, want_w = data[i]
        for rep in range(3):
            Cc = C.copy()
            for M in (A, B, Cc):
                m4ri_amd.pin(M)
            m4ri_amd.mzd_addmul(Cc, A, B, 0)
            assert m4ri_amd.is_pinned(Cc) == 2      # the result lives on the device
            m4ri_amd.sync(Cc)
            assert Cc.equal(want), ("pinned addmul", i, rep)
            m4ri_amd.mzd_addmul(Cc, A, B, 0)        # += again: back to C
            got_w = m4ri_amd.mzd_mul(None, A.window(0, 0, 512, 1024), B.window(0, 0, 1024, 768), 0)
            for M in (A, B, Cc):
                m4ri_amd.unpin(M)
            assert Cc.equal(C) and got_w.equal(want_w), ("chain", i, rep)
    _run_threads([work] * 4)


def test_two_threads_through_the_multi_device_entry_points(oracle):
    """mzd_mul_mp / m4ri_amd_mul_multi from two host threads at once (virtual ranks on this GPU), a third thread on the single-device
    entry points meanwhile: the calls serialise inside the library, every result is the oracle's."""
    m4ri_amd.set_devices([0, 0, 0])
    old = m4ri_amd.set_multi_threshold(512)
    try:
        cases = []
        for i, (m, l, n) in enumerate([(1500, 2000, 1700), (2048, 1024, 2048), (1025, 1025, 1025)]):
            A, B = Mzd.random(m, l, 700 + i), Mzd.random(l, n, 800 + i)
            cases.append((A, B, oracle.mul(None, A, B, 0)))

        def work(i):
            A, B, want = cases[i]
            for rep in range(4):
                if i < 2:
                    assert m4ri_amd.mzd_mul_mp(None, A, B, 0).equal(want), ("mul_mp", i, rep)
                    assert m4ri_amd.mul_multi(Mzd.init(A.nrows, B.ncols), A, B, False, 0, 1 + rep % 2).equal(want), ("mul_multi", i, rep)
                else:
                    assert m4ri_amd.mzd_mul(None, A, B, 0).equal(want), ("mul", i, rep)
        _run_threads([work] * 3)
    finally:
        m4ri_amd.set_multi_threshold(old)


def test_device_api_from_threads_on_their_own_streams(oracle):
    """m4ri_amd_mul_dev from four threads, each on its own HIP stream: the engine has ONE workspace per device, so the products
    are ordered on the device; every result is complete and correct when its stream is synchronised."""
    n = 2048
    w = n // 64
    outs = [None] * 4

    def work(i):
        torch.cuda.set_device(0)
        st = torch.cuda.Stream()
        A = torch.empty((n, w), dtype=torch.int64, device="cuda")
        B, C = torch.empty_like(A), torch.empty_like(A)
        with torch.cuda.stream(st):
            m4ri_amd.fill_dev(A.data_ptr(), w, n, n, 900 + i, st.cuda_stream)
            m4ri_amd.fill_dev(B.data_ptr(), w, n, n, 950 + i, st.cuda_stream)
            for _ in range(5):
                m4ri_amd.mul_dev(C.data_ptr(), w, A.data_ptr(), w, B.data_ptr(), w, n, n, n, False, 256, st.cuda_stream)
        st.synchronize()
        outs[i] = C.cpu().numpy().view(np.uint64)
    _run_threads([work] * 4)
    for i in range(4):
        want = oracle.mul(None, Mzd.random(n, n, 900 + i), Mzd.random(n, n, 950 + i), 0)
        assert np.array_equal(outs[i], want.valid_words()), i
