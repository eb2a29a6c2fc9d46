import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native_lib():
    """Build (if needed) and load libdoda_hip.so; GPU tests must run the native path."""
    from doda_amd.build import build_native
    from doda_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        build_native(verbose=False)
    return _lib.lib()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build_oracle()
    return orc
