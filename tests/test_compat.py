"""The C++ host side above the C ABI: compat/kalmanfilter.h (header-compatible with the reference's
odometry/kalmanfilter.h:21-43) and the headless replay of slam.cpp's loop (compat/replay.cpp)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPLAY = os.path.join(ROOT, "compat", "replay")


@pytest.fixture(scope="module")
def replay_bin():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "2d-ekf-slam_amd", "csrc"), "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "compat"), "-s"])
    return REPLAY


def write_records(path, script):
    with open(path, "w") as f:
        f.write("# dt vel_mm_s rotvel_deg_s compass|nan n fx_mm fy_mm ...\n")
        for st in script:
            comp = "nan" if st["compass"] is None else repr(float(st["compass"]))
            feats = " ".join("%r %r" % (float(fx), float(fy)) for fx, fy in st["feats_mm"])
            f.write("%r %r %r %s %d %s\n" % (st["dt"], st["v"] * 1000.0, st["w"] * 180.0 / 3.141592654, comp, len(st["feats_mm"]), feats))


def test_compat_header_keeps_the_reference_interface():
    src = open(os.path.join(ROOT, "compat", "kalmanfilter.h")).read()
    for decl in ("class KalmanFilter", "double X = 0.0;", "double Y = 0.0;", "double Phi = 0.0;", "int Num_Landmarks = 0;",
                 "void doPropagation(double dt, std::ofstream &covFile, std::ofstream &knownfeaturesFile)",
                 "void doUpdate(Eigen::MatrixXd z_chunk, Eigen::MatrixXd R_chunk)", "void doUpdateCompass(double z, double R)",
                 "#define INF 999999999999", "#define PI 3.141592653589793238462643383279502884197169399375105820974944592307816406286"):
        assert decl in src, decl
    out = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-x", "c++", os.path.join(ROOT, "compat", "kalmanfilter.h")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_replay_fails_loudly_without_a_gpu(replay_bin, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    rec = tmp_path / "rec.txt"
    rec.write_text("0.1 300 2 nan 0\n")
    out = subprocess.run([replay_bin, str(rec), str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 1 and "no HIP device" in out.stderr


def write_state(path, x, P):
    with open(path, "wb") as f:
        np.array([x.size], dtype=np.float64).tofile(f)
        np.ascontiguousarray(x, dtype=np.float64).tofile(f)
        np.ascontiguousarray(P, dtype=np.float64).tofile(f)


def read_state(path):
    a = np.fromfile(path, dtype=np.float64)
    n = int(a[0])
    return a[1:1 + n].copy(), a[1 + n:1 + n + n * n].reshape(n, n).copy()


def synthetic_scan(rng):
    """181 readings of a SICK sweep (slam.cpp:90: 180 degrees): range, local x, local y in mm; some beyond 7 m."""
    ang = np.deg2rad(np.arange(-90, 91))
    rng_mm = rng.uniform(500.0, 9000.0, ang.size)
    return [(float(r), float(r * np.cos(a)), float(r * np.sin(a))) for r, a in zip(rng_mm, ang)]


@pytest.mark.gpu
def test_replay_against_the_oracle_trajectory_and_reference_file_layout(replay_bin, pkg, oc, tmp_path):
    """compat/replay (C++ -> compat/kalmanfilter.h -> C ABI -> HIP) over a config-1 lifecycle, compared with the ORACLE
    driven as slam.cpp:130-204 drives the reference: every gate decision, the pose after every iteration (odomRun.txt),
    the robot covariance corner after every Propagate (covRun.txt), the known-feature lines with the reference's
    stride-1 indexing (kalmanfilter.cpp:56-59), the world-frame feature lines (slam.cpp:173-177), the once-per-second
    scan dump (slam.cpp:184-203), and the final x, P (ekf_get_state).  The files sit where slam.cpp:21-50 puts them,
    which is where plot.py:10-17 and mapping/RealTimePlotting.m:57-102 look."""
    from helpers import assert_state_close
    script = pkg.scenarios.lifecycle_script(steps=150, compass_every=7)
    rng = np.random.default_rng(9)
    scans = {s: synthetic_scan(rng) for s in range(0, len(script), 4)}  # the laser thread delivers a new sweep every 4 iterations
    rec = tmp_path / "rec.txt"
    with open(rec, "w") as f:
        for s, st in enumerate(script):
            if s in scans:
                f.write("scan %d %s\n" % (len(scans[s]), " ".join("%r %r %r" % r for r in scans[s])))
            comp = "nan" if st["compass"] is None else repr(float(st["compass"]))
            feats = " ".join("%r %r" % (float(fx), float(fy)) for fx, fy in st["feats_mm"])
            f.write("%r %r %r %s %d %s\n" % (st["dt"], st["v"] * 1000.0, st["w"] * 180.0 / 3.141592654, comp, len(st["feats_mm"]), feats))
    dump = tmp_path / "final.bin"
    # capacity 4 for a map that ends with 17 landmarks: the reference's state simply grows (Update.cpp:158-177), so the shim
    # must grow its device buffers on the way (ekf_reserve, doubling) -- without a visible effect on any result
    out = subprocess.run([replay_bin, str(rec), str(tmp_path), "4", "--dump-state", str(dump), "--precision", "17"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr

    # the oracle, driven the same way
    x, P = np.zeros(3), np.zeros((3, 3))
    odom, cov, known, feats, decs, scan_pts, chatter = [], [], [], [], [], [], []
    cur_scan, loop_time = [], 0.0
    for s, st in enumerate(script):
        if s in scans:
            cur_scan = scans[s]
        v, w = (st["v"] * 1000.0) / 1000.0, (st["w"] * 180.0 / 3.141592654) * 3.141592654 / 180.0  # kalmanfilter.cpp:19,26
        x, P = oc.propagate(x, P, v, w, oc.make_Q(v), st["dt"])
        cov.append((P[0, 0], P[0, 1], P[1, 0], P[1, 1]))          # kalmanfilter.cpp:51
        n_lm = (x.size - 3) // 2
        for i in range(1, n_lm):                                   # kalmanfilter.cpp:56-59: stride 1, from i = 1
            known.append((x[3 + i], x[4 + i]))
        if st["compass"] is not None:
            chatter.append("Compass:")
            x, P = oc.compass(x, P, st["compass"], 0.0005)
        for fx, fy in st["feats_mm"]:
            z, R = oc.make_measurement(fx, fy)
            x, P, d, m, mh = oc.update(x, P, z.reshape(2, 1), R)
            decs.append((d[0], m[0], mh[0]))
            # what slam.cpp:169-171 and Update.cpp:154,183,191 leave on stdout for this measurement
            chatter.append("Update: %s %d" % ({oc.NEW: "New", oc.OLD: "Old", oc.IGNORE: "Ignore"}[d[0]], (x.size - 3) // 2))
            c, sn = np.cos(x[2]), np.sin(x[2])
            feats.append((z[0] * c - z[1] * sn + x[0], z[0] * sn + z[1] * c + x[1]))  # slam.cpp:173-177
        odom.append((x[0], x[1]))                                  # slam.cpp:181
        loop_time += st["dt"]
        if loop_time > 1.0:                                        # slam.cpp:184-203
            c, sn = np.cos(x[2]), np.sin(x[2])
            for r, lx, ly in cur_scan:
                if r > 7000:
                    continue
                scan_pts.append((lx / 1000.0 * c - ly / 1000.0 * sn + x[0], lx / 1000.0 * sn + ly / 1000.0 * c + x[1]))
            loop_time = 0.0

    def load(rel, cols):
        a = np.loadtxt(str(tmp_path / rel), ndmin=2)
        assert a.shape[1] == cols, rel
        return a

    tol = dict(rtol=1e-6, atol=1e-9)
    dd = load("data/decisionsRun.txt", 3)
    assert [(int(a), int(b)) for a, b in dd[:, :2]] == [(a, b) for a, b, _ in decs]
    assert np.allclose(dd[:, 2], [d[2] for d in decs], **tol)
    assert {d[0] for d in decs} == {oc.NEW, oc.OLD, oc.IGNORE} or {d[0] for d in decs} >= {oc.NEW, oc.OLD}
    assert np.allclose(load("data/odom/odomRun.txt", 2), np.array(odom), **tol)
    assert np.allclose(load("data/cov/covRun.txt", 4), np.array(cov), rtol=1e-6, atol=1e-15)
    assert len(known) > 1000 and np.allclose(load("data/features/knownfeaturesRun.txt", 2), np.array(known), **tol)
    assert np.allclose(load("data/features/featuresRun.txt", 2), np.array(feats), **tol)
    assert len(scan_pts) > 500 and np.allclose(load("data/scan/scanRun.txt", 2), np.array(scan_pts), **tol)
    xg, Pg = read_state(str(dump))
    assert_state_close(xg, Pg, x, P, "replay final state")
    assert np.array_equal(Pg, Pg.T)
    lines = out.stdout.strip().split("\n")
    final = lines[-1].split()
    assert final[0] == "final" and int(final[4]) == (x.size - 3) // 2 and np.allclose([float(v) for v in final[1:4]], x[:3], **tol)
    assert (x.size - 3) // 2 > 16, "the map was meant to outgrow the initial capacity of 4 three times"
    # the reference's only diagnostics, token for token: "Compass: <z>" (slam.cpp:145), "Update: " + "New " / "Old " / "Ignore "
    # (Update.cpp:154,183,191) + Num_Landmarks (slam.cpp:169-171)
    got = [l if l.startswith("Update:") else l.split()[0] for l in lines[:-1]]
    assert got == chatter
    # what plot.py:10-31 does with the three files it opens (relative to the directory the program ran in)
    for rel in ("data/odom/odomRun.txt", "data/features/featuresRun.txt", "data/scan/scanRun.txt"):
        rows = [r for r in open(str(tmp_path / rel)).read().split("\n") if r != ""]
        assert rows and all(len(r.split(" ")) >= 2 and float(r.split(" ")[0]) == float(r.split(" ")[0]) for r in rows)
    assert os.path.isdir(str(tmp_path / "maps"))  # plot.py:8 saves into ./maps/
    # without --precision the files carry what the reference's streams write: their precision is never set (slam.cpp:177,181,200,
    # kalmanfilter.cpp:51,58), i.e. the default 6 significant digits ("%g"): the same run again, every number of every file the "%g" of the
    # round-trip value above -- byte for byte what a reference run prints for these values
    ref_dir = tmp_path / "as_reference"
    ref_dir.mkdir()
    out6 = subprocess.run([replay_bin, str(rec), str(ref_dir), "4", "--quiet"], capture_output=True, text=True)
    assert out6.returncode == 0, out6.stderr
    for rel in ("data/odom/odomRun.txt", "data/cov/covRun.txt", "data/features/featuresRun.txt", "data/features/knownfeaturesRun.txt", "data/scan/scanRun.txt"):
        full = open(str(tmp_path / rel)).read().split("\n")
        six = open(str(ref_dir / rel)).read().split("\n")
        assert len(full) == len(six) and len(full) > 1, rel
        for lf, l6 in zip(full, six):
            assert l6.split(" ") == [("%g" % float(v)) if v else v for v in lf.split(" ")], (rel, lf, l6)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [50, 1024, 4096])
def test_immediate_mode_latency_is_bounded(replay_bin, pkg, tmp_path, pipeline_mode, N):
    """The path slam.cpp really uses: one synchronising call at a time with the public mirrors refreshed after each
    (kalmanfilter.cpp:46-48,85-89).  Round 6: for maps above 256 landmarks the calls are commands to a resident streaming launch
    (N = 1024 / 4096: 62-70 us per 5-call step from C++, p90 77 / 115 us; one launch per call cost 100 / 110 us, p90 185 us); maps of up to
    256 landmarks stream through k_solo (42 us at N = 50).  Measured on idle boxes: C++ 42 / 58 / 65 us per step, Python 100-130 us.  The
    bounds below are tripwires for a regression to one launch per call on a slow or shared host, not the measurement (bench.py's immediate
    leg is): C++ median 200 us, p90 400 us where the calls stream; the Python mirror (ctypes and NumPy conversions) 300 us."""
    M, steps = 4, 80
    x0, P0 = pkg.scenarios.injected_state(N, seed=1, extent=50.0 * (N / 4096.0) ** 0.5)  # constant landmark density
    sc = pkg.scenarios.steady_script(x0, steps=steps, M=M, seed=2, min_separation=1.0)
    bound_us = 200.0
    # C++: compat/replay --timing, starting from the injected state; measurements handed over as robot-frame mm features
    rec = tmp_path / "rec.txt"
    with open(rec, "w") as f:
        for s in range(steps):
            v, w, dt = (float(c) for c in sc["ctrl"][s])
            feats = " ".join("%r %r" % (float(1000.0 * z[0]), float(1000.0 * z[1])) for z in sc["z"][s])
            f.write("%r %r %r nan %d %s\n" % (dt, v * 1000.0, w * 180.0 / 3.141592654, M, feats))
    st = tmp_path / "state.bin"
    write_state(str(st), x0, P0)
    out = subprocess.run([replay_bin, str(rec), str(tmp_path), str(N), "--state", str(st), "--timing"], capture_output=True, text=True)
    assert out.returncode == 0 and "timing" in out.stdout, (out.stdout, out.stderr)
    t = out.stdout.split("timing")[1].split()
    median_cpp, p90_cpp = float(t[3]), float(t[5])
    dd = np.loadtxt(str(tmp_path / "data" / "decisionsRun.txt"), ndmin=2)
    assert dd.shape[0] == steps * M and np.all(dd[:, 0] == 2)  # every update matched an old landmark: steady state
    assert [int(m) for m in dd[:, 1]] == [3 + 2 * int(t_) for t_ in sc["target"].ravel()]
    # Python mirror
    import time
    kf = pkg.KalmanFilter(capacity_landmarks=N)
    try:
        kf.set_state(x0, P0)
        per_step = []
        for s in range(steps):
            v, w, dt = sc["ctrl"][s]
            t0 = time.perf_counter()
            kf.doPropagation(dt, v * 1000.0, w * 180.0 / 3.141592654)
            for m in range(M):
                kf.doUpdate(sc["z"][s, m].reshape(2, 1), sc["R"][s, m].reshape(2, 2, order="F"))
            per_step.append((time.perf_counter() - t0) * 1e6)
    finally:
        kf._f.close()  # (a failing assertion below must not leave the handle's chain workgroups claimed for the rest of the suite)
    median_py = float(np.median(per_step[16:]))
    print("immediate mode N=%d: %.0f us/step C++ (p90 %.0f), %.0f us/step Python" % (N, median_cpp, p90_cpp, median_py))
    assert median_cpp < bound_us and median_py < 300.0, (N, median_cpp, median_py)
    if N > 256:
        assert "streaming 1" in out.stdout, out.stdout[-300:]
        # (the p90 is the step behind a full window: with the pass forced in place -- this suite's "inplace" mode overrides the shim's choice
        # of the overlapped pipeline from 2048 landmarks on -- that step waits 100 us for the pass at N = 4096)
        # (p90 bound 400 us: 77-115 us on an idle box; the step behind a full window carries five host-side HIP calls -- event, stream wait,
        # pass, mark, the next streaming launch -- and a busy host stretches those, 173 us seen once on a shared pod)
        if not (N >= 2048 and pipeline_mode == "inplace"):
            assert p90_cpp < 400.0, (N, p90_cpp, out.stdout[-300:])


def test_compat_featuredetector_header_keeps_the_reference_interface():
    """compat/featuredetector.h against perception/featuredetector.h:60-72: the class, NO_COMPASS, getFeatures' signature
    and the Feature struct's fields; and it compiles against the stand-ins."""
    src = open(os.path.join(ROOT, "compat", "featuredetector.h")).read()
    for decl in ("class FeatureDetector", "struct Feature", "NO_COMPASS", "FeatureDetector(ArSick *",
                 "int getFeatures(std::vector<Feature> *featVec, double *structCompass, double curPhi)",
                 # the public constants of featuredetector.h:27-36
                 "static const int MAX_DIST = 8000;", "static const int MIN_DIST = 1000 * 1000;", "static const int MIN_POINTS = 3;",
                 "static const int POINT_DIST = 600;", "double CORNER_THETA = 22.0 * 3.141592654 / 180.0;", "static const int CORNER_DIST = 90000;",
                 "double COMPASS_THRESH = 10 * 3.141592654 / 180.0;"):
        assert decl in src, decl
    # ... and they are what the kernels use
    dev = open(os.path.join(ROOT, "2d-ekf-slam_amd", "csrc", "feat_device.h")).read()
    for d in ("#define FEAT_MAX_DIST 8000", "#define FEAT_MIN_DIST (1000 * 1000)", "#define FEAT_MIN_POINTS 3", "#define FEAT_POINT_DIST 600", "#define FEAT_CORNER_DIST 90000"):
        assert d in dev, d
    out = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "compat", "standin"),
                          "-x", "c++", os.path.join(ROOT, "compat", "featuredetector.h")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REFERENCE, "slam.cpp")), reason="the reference tree is not on this machine (it never travels to the GPU box)")
def test_the_reference_slam_cpp_type_checks_unchanged_against_the_compat_headers(tmp_path):
    """north_star: "so slam.cpp ... feed it unchanged".  The compiler's front end (g++ -fsyntax-only: nothing is built, linked or run)
    goes over the reference's own slam.cpp -- reached through a symbolic link, so that its quoted includes resolve in a directory of
    ours -- with compat/kalmanfilter.h at odometry/kalmanfilter.h and compat/featuredetector.h at features/featuredetector.h
    (slam.cpp:11,13), the two declaration-only test doubles at Aria.h and Eigen/Dense (slam.cpp:5,10; this image has neither), and the
    reference's own movement/movementcontroller.h and features/houghtransform.h (slam.cpp:12,14: declarations slam.cpp needs, out of
    this path's scope) where they are.  Every use slam.cpp makes of KalmanFilter and FeatureDetector -- the constructors :110,:127, doPropagation
    :136, getFeatures :141, NO_COMPASS :144, doUpdateCompass :146, doUpdate with Eigen matrices :170, the public mirrors :171-181 --
    must type-check, or the drop-in claim is false."""
    tu = tmp_path / "tu"
    for d in ("odometry", "features", "movement", "Eigen"):
        (tu / d).mkdir(parents=True)
    os.symlink(os.path.join(REFERENCE, "slam.cpp"), tu / "slam.cpp")
    (tu / "odometry" / "kalmanfilter.h").write_text('#include "%s"\n' % os.path.join(ROOT, "compat", "kalmanfilter.h"))
    (tu / "features" / "featuredetector.h").write_text('#include "%s"\n' % os.path.join(ROOT, "compat", "featuredetector.h"))
    (tu / "Aria.h").write_text('#include "%s"\n' % os.path.join(ROOT, "compat", "standin", "aria_standin.h"))
    (tu / "Eigen" / "Dense").write_text('#include "%s"\n' % os.path.join(ROOT, "compat", "standin", "eigen_standin.h"))
    os.symlink(os.path.join(REFERENCE, "movement", "movementcontroller.h"), tu / "movement" / "movementcontroller.h")
    os.symlink(os.path.join(REFERENCE, "features", "houghtransform.h"), tu / "features" / "houghtransform.h")
    out = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-w", "-I", str(tu), str(tu / "slam.cpp")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    # the check bites: without doUpdateCompass the same translation unit must fail
    broken = open(os.path.join(ROOT, "compat", "kalmanfilter.h")).read().replace("void doUpdateCompass(", "void doUpdateCompass_gone(")
    (tu / "odometry" / "kalmanfilter.h").write_text(broken.replace('"../include/', '"%s/' % os.path.join(ROOT, "include")).replace('"standin/', '"%s/' % os.path.join(ROOT, "compat", "standin")))
    out = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-w", "-I", str(tu), str(tu / "slam.cpp")], capture_output=True, text=True)
    assert out.returncode != 0 and "doUpdateCompass" in out.stderr


@pytest.mark.gpu
def test_replay_detect_runs_perception_and_filter_end_to_end_like_the_oracle(replay_bin, pkg, oc, tmp_path):
    """The whole slam.cpp:130-204 loop with perception in it: every iteration a new sweep -> FeatureDetector::getFeatures
    (compat/featuredetector.h -> feat_extract on the GPU) -> doUpdateCompass when a wall gives a heading
    (slam.cpp:144-147) -> one doUpdate per corner (:152-170).  The oracle side runs the SAME loop with the CPU restatement
    of the detector (oracle/features_oracle.c) feeding the CPU restatement of the filter (oracle/ekf_oracle.c): the
    compass values, every gate decision, the trajectory and the final x, P must agree, and the filter must end close to
    the simulated truth."""
    from helpers import assert_state_close
    from oracle import features_c as fc
    drive = pkg.scenarios.simulated_drive(steps=120)
    rec = tmp_path / "rec.txt"
    with open(rec, "w") as f:
        for it in drive:
            r, lx, ly = it["scan"]
            f.write("scan %d %s\n" % (r.size, " ".join("%r %r %r" % (float(a), float(b), float(c)) for a, b, c in zip(r, lx, ly))))
            f.write("%r %r %r nan 0\n" % (float(it["dt"]), float(it["v_mm_s"]), float(it["rot_deg_s"])))
    dump = tmp_path / "final.bin"
    out = subprocess.run([replay_bin, str(rec), str(tmp_path), "64", "--detect", "--quiet", "--dump-state", str(dump), "--precision", "17"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr

    x, P = np.zeros(3), np.zeros((3, 3))
    offset = np.array([100.0])  # featuredetector.h:71: COMPASS_OFFSET starts at NO_COMPASS
    comps, decs, odom = [], [], []
    for it in drive:
        v, w = it["v_mm_s"] / 1000.0, it["rot_deg_s"] * 3.141592654 / 180.0
        x, P = oc.propagate(x, P, v, w, oc.make_Q(v), it["dt"])
        o = fc.extract(*it["scan"], max_corners=64)
        cp = fc.compass(o["lines"], x[2], offset)       # featuredetector.cpp:61 with curPhi = ekf->Phi after Propagate
        if cp != 100.0:
            comps.append(cp)
            x, P = oc.compass(x, P, cp, 0.0005)
        for cx, cy in o["corners"]:
            z, R = oc.make_measurement(cx, cy)
            x, P, d, m, mh = oc.update(x, P, z.reshape(2, 1), R)
            decs.append((d[0], m[0], mh[0]))
        odom.append((x[0], x[1]))
    tol = dict(rtol=1e-6, atol=1e-9)
    assert len(comps) > 100 and np.allclose(np.loadtxt(str(tmp_path / "data/compassRun.txt")), comps, **tol)
    dd = np.loadtxt(str(tmp_path / "data/decisionsRun.txt"), ndmin=2)
    assert len(decs) > 80 and [(int(a), int(b)) for a, b in dd[:, :2]] == [(a, b) for a, b, _ in decs]
    assert np.allclose(dd[:, 2], [d[2] for d in decs], **tol)
    assert {oc.NEW, oc.OLD} <= {d[0] for d in decs}
    assert np.allclose(np.loadtxt(str(tmp_path / "data/odom/odomRun.txt"), ndmin=2), np.array(odom), **tol)
    xg, Pg = read_state(str(dump))
    assert_state_close(xg, Pg, x, P, "replay --detect final state")
    # and the estimate is right: the truth, expressed in the frame the filter started in
    p0 = np.array([-3500.0, -2500.0, 0.35])
    d = (drive[-1]["truth"][:2] - p0[:2]) / 1000.0
    c, s = np.cos(p0[2]), np.sin(p0[2])
    truth = np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1], drive[-1]["truth"][2] - p0[2]])
    assert np.all(np.abs(xg[:3] - truth) < [0.1, 0.1, 0.02]), (xg[:3], truth)
