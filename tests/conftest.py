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
