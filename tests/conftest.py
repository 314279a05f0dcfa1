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


def pytest_generate_tests(metafunc):
    # every GPU test runs in both pipeline modes
    if metafunc.definition.get_closest_marker("gpu") is not None and "pipeline_mode" in metafunc.fixturenames:
        metafunc.parametrize("pipeline_mode", ["inplace", "overlap"], indirect=True)


@pytest.fixture(autouse=True)
def pipeline_mode(request, monkeypatch):
    """GPU tests run twice: dense pass in place between the windows, and dense pass overlapped with the next
    window's chain kernels (ekf_params.overlap; the environment variable overrides the parameter)."""
    mode = getattr(request, "param", None)
    if mode is not None:
        monkeypatch.setenv("EKF_OVERLAP", "1" if mode == "overlap" else "0")
    return mode


@pytest.fixture(autouse=True)
def hang_watchdog(request):
    """EKF_TEST_WATCHDOG=<seconds>: a test stuck inside a native call dumps all Python stacks to stderr and exits
    (pytest-timeout cannot interrupt a thread that is blocked in C)."""
    secs = os.environ.get("EKF_TEST_WATCHDOG")
    if not secs:
        yield
        return
    import faulthandler
    sys.stderr.write("[test] %s\n" % request.node.nodeid)
    sys.stderr.flush()
    faulthandler.dump_traceback_later(float(secs), exit=True)
    yield
    faulthandler.cancel_dump_traceback_later()
