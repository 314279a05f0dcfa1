"""The diagnostic builds of the library take part in the suite.  k_chain's correctness rests on a register-allocator workaround
(-mllvm -vgpr-regalloc=basic, 2d-ekf-slam_amd/csrc/Makefile) that was found through out-of-bounds stores; the bounds-checking variant
(make check: -DEKF_CHAIN_CHECK, every data-dependent global index of k_chain range-checked on the device, first violation kept in
dv.dbg[8..11]) therefore runs randomised API traffic, large multi-workgroup maps and lifecycles with New landmarks in a child
process and must report no violation; the build itself refuses a compiler the assembly lint was not validated for."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "2d-ekf-slam_amd", "lib")

_CHILD = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    sys.path.insert(0, os.path.join(%r, "tests"))
    import pytest
    rc = pytest.main(["-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(%r, "tests", "test_gpu_parity.py"), "-k",
                      "random_operation_sequences_vs_oracle or lifecycle_with_multi_workgroup_capacity or scripted_lifecycle_with_new_landmarks "
                      "or small_batch_of_multi_workgroup_filters or several_landmarks_per_worker_thread or random_scripted_pieces_on_several_workgroups"])
    import __graft_entry__ as ge
    pkg = ge.load_package()
    assert os.path.basename(pkg.ekfslam.LIB_PATH) == "libekfslam_hip_check.so", pkg.ekfslam.LIB_PATH
    L = pkg.load()
    L.ekf_debug_check_violations.restype = __import__("ctypes").c_long
    v = L.ekf_debug_check_violations()
    print("CHECKED rc=%%d violations=%%d" %% (int(rc), v))
    sys.exit(int(rc) or (3 if v else 0))
""") % (ROOT, ROOT, ROOT)


@pytest.mark.gpu
def test_bounds_checking_library_sees_no_violation(pipeline_mode):
    """About 200 runs of the parity suite -- 26 seeds of random API traffic (maps of up to 500 landmarks on 3 to 8 workgroups),
    60 seeds of random scripted pieces on 2 to 4 workgroups, lifecycles with New landmarks at five window lengths, landmarks beyond
    the register-resident one per thread -- on libekfslam_hip_check.so, k_chain forced for the small maps too (EKF_SOLO=0: the
    checks live in k_chain).  Every test passes as on the product library and no handle saw an index out of range."""
    if pipeline_mode != "inplace":
        pytest.skip("once is enough: the child runs every selected test in both pipeline modes itself")
    lib = os.path.join(LIBDIR, "libekfslam_hip_check.so")
    assert os.path.exists(lib), "build it: make -C 2d-ekf-slam_amd/csrc check (part of `make all`)"
    env = {k: v for k, v in os.environ.items() if not k.startswith("EKF")}
    env.update(EKFSLAM_LIB=lib, EKF_SOLO="0", EKF_TEST_FEWER_SEEDS="1")
    r = subprocess.run([sys.executable, "-c", _CHILD], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    tail = (r.stdout[-1500:], r.stderr[-1500:])
    assert "CHECKED rc=0 violations=0" in r.stdout and r.returncode == 0, tail
    assert "EKF_CHAIN_CHECK:" not in r.stderr, tail


def test_build_records_the_compiler_it_was_validated_for():
    """`make all` writes the compiler's version line beside the lint's verdict and stops on another compiler (the exec-mask lint
    and the basic register allocator were examined for exactly one)."""
    mk = open(os.path.join(ROOT, "2d-ekf-slam_amd", "csrc", "Makefile")).read()
    assert "VALIDATED_COMPILER :=" in mk and "libekfslam_hip_check.so" in mk.split("all:")[1].split("\n")[0]
    lint = os.path.join(LIBDIR, "asm", "lint.ok")
    if not os.path.exists(lint):
        pytest.skip("library not built here")
    text = open(lint).read()
    assert "0 suspicious site(s)" in text
    validated = [l for l in mk.split("\n") if l.startswith("VALIDATED_COMPILER :=")][0].split(":=", 1)[1].strip()
    assert "compiler: " + validated in text
