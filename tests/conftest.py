import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle (test infrastructure); built on demand from oracle/Makefile."""
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def native_lib():
    """libgendr_hip.so; built on demand (hipcc cross-compiles gfx950 without a GPU)."""
    from gendr_amd import build, _native
    build.build_all()
    return _native.lib()
