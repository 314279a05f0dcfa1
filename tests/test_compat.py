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
                 "void doUpdate(Eigen::MatrixXd z_chunk, Eigen::MatrixXd R_chunk)", "void doUpdateCompass(double z, double R)"):
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


@pytest.mark.gpu
def test_replay_matches_python_mirror_and_file_formats(replay_bin, pkg, tmp_path):
    script = pkg.scenarios.lifecycle_script(steps=120, compass_every=7)
    rec = tmp_path / "rec.txt"
    write_records(str(rec), script)
    out = subprocess.run([replay_bin, str(rec), str(tmp_path), "64"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    final = [float(v) for v in out.stdout.split()[1:4]] + [int(out.stdout.split()[4])]

    kf = pkg.KalmanFilter(capacity_landmarks=64)
    odom, decs = [], []
    for st in script:
        kf.doPropagation(st["dt"], st["v"] * 1000.0, st["w"] * 180.0 / 3.141592654)
        if st["compass"] is not None:
            kf.doUpdateCompass(st["compass"], 0.0005)
        for fx, fy in st["feats_mm"]:
            z, R = pkg.scenarios.measurement_from_feature_mm(fx, fy)
            kf.doUpdate(z.reshape(2, 1), R)
            decs.append(kf.last_decisions[0][:2])
        odom.append((kf.X, kf.Y))
    assert final[3] == kf.Num_Landmarks
    assert np.allclose(final[:3], [kf.X, kf.Y, kf.Phi], rtol=0, atol=1e-12)
    # odomRun.txt: "X Y" per loop iteration (slam.cpp:181)
    od = np.loadtxt(str(tmp_path / "odomRun.txt"))
    assert od.shape == (len(script), 2) and np.allclose(od, np.array(odom), rtol=0, atol=1e-12)
    # covRun.txt: "P00 P01 P10 P11" per propagate (kalmanfilter.cpp:51)
    cov = np.loadtxt(str(tmp_path / "covRun.txt"))
    assert cov.shape == (len(script), 4) and np.allclose(cov[:, 1], cov[:, 2])
    # featuresRun.txt: one "x y" world-frame line per feature (slam.cpp:177)
    nfeat = sum(len(st["feats_mm"]) for st in script)
    assert np.loadtxt(str(tmp_path / "featuresRun.txt")).shape == (nfeat, 2)
    dd = np.loadtxt(str(tmp_path / "decisionsRun.txt"))
    assert [(int(a), int(b)) for a, b in dd[:, :2]] == decs
