import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "deepsphere-weather_amd")
for p in (PKG, REPO, os.path.join(REPO, "tests", "golden"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")



def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """Build libdsw_hip.so if hipcc is around and the in-tree .so is stale/missing."""
    from dsw_amd import build as dsw_build

    if dsw_build.needs_build():
        dsw_build.build(verbose=False)
    yield


@pytest.fixture()
def oracle_backend():
    """Route CPU tensors of the host logic through the oracle (checker), restore strict mode after."""
    from dsw_amd import functional
    from _oracle_backend import OracleBackend

    functional.set_test_backend(OracleBackend())
    yield
    functional.set_test_backend(None)


def load_golden(name):
    import numpy as np

    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
