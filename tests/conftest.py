import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU case")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build libhadi.so (hipcc cross-compiles without a GPU) and the oracle once per session."""
    import __graft_entry__ as G
    G.build_libhadi()
    G.build_oracle()
    yield


@pytest.fixture(scope="session")
def solver():
    import pde_based_heston_solver_gpu_accelerated_amd as H
    s = H.HestonADI(0)
    yield s
    s.close()
