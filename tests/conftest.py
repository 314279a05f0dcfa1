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


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """One loud line whenever the Eigen-typed restatement (oracle/ekf_oracle_eigen.cpp: JacobiSVD, dynamic inverse(), Eigen's product
    order -- the code paths the reference really runs, Update.cpp:127-136,186-188) could not be built: the C oracle is "parity
    unpinned", and the first box with <Eigen/Dense> should turn that file from never-compiled into evidence, not skip silently."""
    try:
        from oracle import oracle_c
        have = oracle_c.eigen_lib() is not None
    except Exception as e:  # (a broken oracle build is the tests' business, not the summary's)
        terminalreporter.write_line("EIGEN RESTATEMENT NOT RUN: the oracle could not be loaded (%s)" % e, yellow=True, bold=True)
        return
    if have:
        terminalreporter.write_line("EIGEN RESTATEMENT BUILT: tests/test_oracle_eigen.py compares the C oracle with Eigen's own JacobiSVD / inverse() / products", green=True)
    else:
        terminalreporter.write_line("EIGEN RESTATEMENT NOT RUN: <Eigen/Dense> is not installed here, oracle/ekf_oracle_eigen.cpp was not compiled and "
                                    "tests/test_oracle_eigen.py skipped -- the oracle stays parity-unpinned (EIGEN_INC=-I/path make -C oracle enables it)",
                                    yellow=True, bold=True)


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
