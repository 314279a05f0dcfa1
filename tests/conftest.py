import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def oc():
    from oracle import oracle_c
    oracle_c.build()
    return oracle_c


@pytest.fixture(scope="session")
def npo():
    from oracle import ekf_numpy
    return ekf_numpy
