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


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """when oracle/_ref is missing on the GPU box the reference-pin tests SKIP and the run stays green; the
    count of launches of the reference's own kernels goes into the tail of the output so that the driver's record shows it."""
    if 'parity' in sys.modules and config.getoption('-m') and 'not gpu' not in config.getoption('-m'):
        import parity
        from oracle import ref_gpu
        terminalreporter.write_line('ref_pin: ran %d launches of the reference kernels (oracle/_ref %s)'
                                    % (parity.REF_PIN_CALLS, 'present' if ref_gpu.available('render') else 'MISSING: pin tests skipped'))
