"""pytest configuration: the `gpu` marker and the CPU-side checkers (oracle + reference build)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import cpu_libs
    return cpu_libs.oracle()


@pytest.fixture(scope="session")
def reference():
    import cpu_libs
    ref = cpu_libs.reference()
    if ref is None:
        pytest.skip("oracle/_ref/libm4ri_ref.so not built (needs /root/reference: `make -C oracle ref`)")
    return ref


@pytest.fixture(scope="session", autouse=True)
def _every_product_on_the_gpu():
    """The parity tests are about the HIP path: in this process every product goes to the GPU, however small.  (The library's host
    routine for tiny products has its own tests, test_small_products.py, which switch its threshold back on; programs the tests
    start -- bench.py, the LD_PRELOAD driver -- run with the library's default.)"""
    import m4ri_amd
    old = m4ri_amd.set_small_product_threshold(0)
    yield
    m4ri_amd.set_small_product_threshold(old)
