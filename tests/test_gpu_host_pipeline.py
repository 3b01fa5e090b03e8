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


@pytest.mark.parametrize("m,l,n", [(16384, 640, 8192), (32768, 512, 16384), (2048, 512, 40000)])
def test_a_reused_result_block_is_written_completely(oracle, m, l, n):
    """mzd_mul(NULL, ...) takes a parked block of the same size back "as it is" (mzd_api.hip: late_begin) on the grounds that every
    run path overwrites every valid word.  The check behind that sentence: in a child process with M4RI_AMD_POISON_RESULT=1 the
    reused block is filled with a pattern first; the pipelined and the one-shot path must still return the oracle's bits (and
    zero padding words)."""
    import os
    import subprocess
    import sys
    code = f'''
import sys, numpy as np
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}); sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})
import m4ri_amd, cpu_libs
from m4ri_amd.mzd import Mzd
m4ri_amd.init(0)
orc = cpu_libs.oracle()
lib = m4ri_amd.lib()
for pipeline in (1, 0):
    m4ri_amd.set_host_pipeline(pipeline)
    for rep in range(3):
        A, B = Mzd.random({m}, {l}, 10 + rep), Mzd.random({l}, {n}, 20 + rep)
        want = orc.mul(None, A, B, 0)
        r = lib.mzd_mul(None, A.ptr, B.ptr, 0)          # the library's own allocator (no libm4ri in this process): fresh, then reused blocks
        from m4ri_amd.mzd import from_struct_ptr
        s = r.contents
        raw = np.ctypeslib.as_array(s.data, shape=(s.nrows * s.rowstride,)).reshape(s.nrows, s.rowstride)
        assert not raw[:, s.width:].any(), ("padding", pipeline, rep)
        assert from_struct_ptr(r).equal(want), ("bits", pipeline, rep)
        lib.m4ri_amd_result_free(r)                     # parks the block: the next product of this size takes it back
print("poisoned blocks OK")
'''
    env = dict(os.environ, M4RI_AMD_POISON_RESULT="1", M4RI_AMD_SMALL_THRESHOLD="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "poisoned blocks OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def _child(code, env_extra, timeout=900):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pre = f"import sys, json, hashlib, numpy as np\nsys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})\n"
    r = subprocess.run([sys.executable, "-c", pre + code], capture_output=True, text=True, env=dict(os.environ, **env_extra), timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout, r.stderr


@pytest.mark.parametrize("w7", ["1", "0"])
def test_winograd_at_the_top_of_the_block_pipeline(w7):
    """The 2 x 2 x 2 block grid of the host pipeline run as Strassen-Winograd (7 block products in the order their quadrants arrive,
    the quadrants of C accumulated on the device; mzd_api.hip run_pipelined) against the oracle, at sizes the oracle finishes in
    seconds (the grid forced with M4RI_AMD_PIPE_GRID; by itself it is chosen from 65536^3 on; the schedule is opt-in, M4RI_AMD_PIPE_W7=1:
    it measured slower than the 8 classical block products, which is what =0 and the default run): same bits.  Unequal halves and C += A*B keep the classical grid; C == NULL and C given both covered."""
    code = '''
import m4ri_amd, cpu_libs
from m4ri_amd.mzd import Mzd
m4ri_amd.init(0); m4ri_amd.set_host_pipeline(1); m4ri_amd.set_small_product_threshold(0)
orc = cpu_libs.oracle()
for (m, l, n) in [(8192, 2048, 2048), (16384, 4096, 8192), (8192, 1280, 1152), (8192 + 4096, 2048, 2048), (8192, 2048 + 64, 2048)]:
    A, B = Mzd.random(m, l, 1), Mzd.random(l, n, 2)
    want = orc.mul(None, A, B, 0)
    assert m4ri_amd.mzd_mul(None, A, B, 0).equal(want), (m, l, n)
    C = Mzd.random(m, n, 3)
    assert m4ri_amd.mzd_mul(C, A, B, 0).equal(want), (m, l, n)
    C0 = Mzd.random(m, n, 4)
    assert m4ri_amd.mzd_addmul(C0.copy(), A, B, 0).equal(orc.addmul(C0.copy(), A, B, 0)), (m, l, n)
print("ok")
'''
    out, err = _child(code, {"M4RI_AMD_PIPE_GRID": "2,2,2", "M4RI_AMD_PIPE_W7": w7, "M4RI_AMD_PIPE_TRACE": "1"})
    assert "ok" in out
    assert ("Strassen-Winograd at the top" in err) == (w7 == "1")          # the schedule really ran (equal halves), and the switch switches it off
    assert "grid 2 x 2 x 2:" in err or w7 == "1"                          # unequal halves / addmul: the classical grid


@pytest.mark.parametrize("w7", ["0", "1"])
def test_config3_through_the_host_entry_point_matches_the_reference_sha256(w7):
    """BASELINE.json configs[2] the way the reference's own bench calls it -- mzd_mul(NULL, A, B, 0) on host matrices, 65536^3, seeds
    3, 4 -- through the block pipeline (the default 8 classical block products, and the optional Strassen-Winograd top level): the
    SHA-256 of the real reference's product."""
    code = '''
import m4ri_amd
from m4ri_amd.mzd import Mzd
m4ri_amd.init(0)
n = 65536
A, B = Mzd.random(n, n, 3), Mzd.random(n, n, 4)
C = m4ri_amd.mzd_mul(None, A, B, 0)
print("sha", hashlib.sha256(C.masked().tobytes()).hexdigest())
'''
    import json
    import os
    out, err = _child(code, {"M4RI_AMD_PIPE_TRACE": "1", "M4RI_AMD_PIPE_W7": w7}, timeout=1500)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = [e["sha256"] for e in json.load(open(os.path.join(root, "tests", "golden", "sha256.json")))
            if (e["op"], e["m"], e["l"], e["n"], e["seed_a"], e["seed_b"]) == ("mul", 65536, 65536, 65536, 3, 4) and not e.get("cutoff")]
    assert want and ("sha " + want[0]) in out and ("Strassen-Winograd at the top" in err) == (w7 == "1")
