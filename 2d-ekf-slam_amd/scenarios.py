"""Synthetic workloads for the EKF hot path.  Pure NumPy, deterministic per seed.

These build *inputs* only (states to inject, odometry, range/bearing measurements converted the
way slam.cpp:152-167 converts them).  No filter arithmetic lives here.
"""
import math

import numpy as np

SIGMA_V, SIGMA_W = 0.01, 0.04          # kalmanfilter.cpp:28-29
VAR_RANGE, VAR_BEARING = 0.0025, 0.0001  # slam.cpp:165
COMPASS_VAR = 0.0005                   # slam.cpp:146


def measurement_from_feature_mm(fx_mm, fy_mm):
    """slam.cpp:158-167: z in metres and R = G diag(0.0025,0.0001) G^T (2x2 matrix)."""
    fx, fy = fx_mm / 1000.0, fy_mm / 1000.0
    dist = math.sqrt(fx * fx + fy * fy)
    b = math.atan2(fy, fx)
    G = np.array([[math.cos(b), -dist * math.sin(b)], [math.sin(b), dist * math.cos(b)]])
    return np.array([fx, fy]), G @ np.diag([VAR_RANGE, VAR_BEARING]) @ G.T


def injected_state(n_landmarks, seed, extent=50.0, rank=8):
    """Configs 2-4: landmarks uniform in [-extent, extent]^2, robot at the origin with phi = 0.3,
    P = D + U U^T (D = diag U(0.01,0.02), U n x rank ~ N(0, 1e-3^2)), bitwise symmetric."""
    rng = np.random.default_rng(seed)
    n = 3 + 2 * n_landmarks
    x = np.empty(n)
    x[0:3] = (0.0, 0.0, 0.3)
    x[3:] = rng.uniform(-extent, extent, size=2 * n_landmarks)
    d = rng.uniform(0.01, 0.02, size=n)
    U = rng.normal(0.0, 1e-3, size=(n, rank))
    P = U @ U.T
    P[np.diag_indices(n)] += d
    P = 0.5 * (P + P.T)
    return x, P


def steady_script(x0, steps, M, seed, v=0.3, w=0.05, dt=0.05, rmin=2.0, rmax=9.0, noise_scale=0.5,
                  min_separation=1.5):
    """Configs 2-4 step script: every step is one propagate (v, w, dt) followed by M single
    measurements, each aimed at a distinct already-mapped landmark rmin..rmax metres away whose
    nearest other landmark is at least min_separation away, so that the reference's gate takes
    the Old branch with margin.  The 'truth' world is the injected estimate itself.

    Returns dict(ctrl (steps,3), z (steps,M,2), R (steps,M,4 column-major), target (steps,M)
    0-based landmark ids, truth (steps,3) pose after each propagate)."""
    rng = np.random.default_rng(seed)
    L = x0[3:].reshape(-1, 2)
    pose = np.array(x0[0:3], dtype=np.float64)
    ctrl = np.tile(np.array([v, w, dt]), (steps, 1))
    z = np.empty((steps, M, 2))
    R = np.empty((steps, M, 4))
    target = np.empty((steps, M), dtype=np.int64)
    truth = np.empty((steps, 3))
    # isolation of each landmark (distance to its nearest neighbour), via a coarse grid
    iso = _nearest_neighbour_distance(L)
    for s in range(steps):
        pose = pose + dt * np.array([v * math.cos(pose[2]), v * math.sin(pose[2]), w])
        truth[s] = pose
        d = L - pose[0:2]
        r = np.hypot(d[:, 0], d[:, 1])
        ok = np.flatnonzero((r >= rmin) & (r <= rmax) & (iso >= min_separation))
        if ok.size < M:  # sparse maps: fall back to the closest well-separated landmarks beyond rmin
            cand = np.flatnonzero((r >= rmin) & (iso >= min_separation))
            if cand.size < M:
                cand = np.flatnonzero(r >= rmin)
            ok = cand[np.argsort(r[cand])[:max(M, 1)]]
        pick = rng.choice(ok, size=M, replace=False)
        c, sn = math.cos(pose[2]), math.sin(pose[2])
        for m, li in enumerate(pick):
            rel = np.array([c * d[li, 0] + sn * d[li, 1], -sn * d[li, 0] + c * d[li, 1]])  # C^T (pL - pR)
            rr = math.hypot(rel[0], rel[1]) + noise_scale * math.sqrt(VAR_RANGE) * rng.standard_normal()
            bb = math.atan2(rel[1], rel[0]) + noise_scale * math.sqrt(VAR_BEARING) * rng.standard_normal()
            zz, RR = measurement_from_feature_mm(1000.0 * rr * math.cos(bb), 1000.0 * rr * math.sin(bb))
            z[s, m] = zz
            R[s, m] = RR.ravel(order="F")
            target[s, m] = li
    return dict(ctrl=ctrl, z=z, R=R, target=target, truth=truth)


def _nearest_neighbour_distance(L):
    n = L.shape[0]
    if n <= 2048:
        d2 = ((L[:, None, :] - L[None, :, :]) ** 2).sum(-1)
        d2[np.diag_indices(n)] = np.inf
        return np.sqrt(d2.min(axis=1))
    out = np.empty(n)
    for a in range(0, n, 1024):
        d2 = ((L[a:a + 1024, None, :] - L[None, :, :]) ** 2).sum(-1)
        d2[np.arange(d2.shape[0]), np.arange(a, a + d2.shape[0])] = np.inf
        out[a:a + 1024] = np.sqrt(d2.min(axis=1))
    return out


def lifecycle_script(seed=20260001, n_landmarks=50, steps=1000, v=0.3, radius=8.0, dt=0.1, max_range=8.0,
                     max_feats=4, compass_every=0):
    """Config 1: full lifecycle from x = 0_3, P = 0 (kalmanfilter.cpp:10-11).  A robot drives a
    circle of the given radius; landmarks are uniform in the annulus 3..12 m around the circle's
    centre; the sensor sees landmarks within max_range and +-90 degrees (SICK: slam.cpp:90,
    houghtransform.h:20), at most max_feats nearest per step.  Odometry noise per
    kalmanfilter.cpp:28-37, range/bearing noise per slam.cpp:165.

    Returns a list of steps: dict(v, w, dt, feats_mm [(fx,fy),...], compass or None, truth pose)."""
    rng = np.random.default_rng(seed)
    w = v / radius
    ang = rng.uniform(0, 2 * math.pi, n_landmarks)
    rad = np.sqrt(rng.uniform(3.0 ** 2, 12.0 ** 2, n_landmarks))
    # robot starts at the origin heading +x; circle centre is at (0, radius)
    L = np.stack([rad * np.cos(ang), radius + rad * np.sin(ang)], axis=1)
    pose = np.zeros(3)
    out = []
    for s in range(steps):
        pose = pose + dt * np.array([v * math.cos(pose[2]), v * math.sin(pose[2]), w])
        v_meas = v + SIGMA_V * v * rng.standard_normal()
        w_meas = w + SIGMA_W * v * rng.standard_normal()
        d = L - pose[0:2]
        c, sn = math.cos(pose[2]), math.sin(pose[2])
        rel = np.stack([c * d[:, 0] + sn * d[:, 1], -sn * d[:, 0] + c * d[:, 1]], axis=1)
        r = np.hypot(rel[:, 0], rel[:, 1])
        b = np.arctan2(rel[:, 1], rel[:, 0])
        vis = np.flatnonzero((r <= max_range) & (np.abs(b) <= math.pi / 2) & (r > 0.3))
        vis = vis[np.argsort(r[vis])[:max_feats]]
        feats = []
        for li in vis:
            rr = r[li] + math.sqrt(VAR_RANGE) * rng.standard_normal()
            bb = b[li] + math.sqrt(VAR_BEARING) * rng.standard_normal()
            feats.append((1000.0 * rr * math.cos(bb), 1000.0 * rr * math.sin(bb)))
        comp = None
        if compass_every and (s % compass_every) == compass_every - 1:
            comp = (pose[2] + math.sqrt(COMPASS_VAR) * rng.standard_normal()) % (2 * math.pi)
        out.append(dict(v=v_meas, w=w_meas, dt=dt, feats_mm=feats, compass=comp, truth=pose.copy()))
    return out


def simulated_scan(seed, n_rays=181, noise_mm=8.0, max_range_mm=12000.0):
    """One sweep of a SICK LMS-200 as the reference reads it (slam.cpp:90: 180 degrees; featuredetector.cpp:18-22:
    range, local x, local y per reading, mm): the robot stands in a random rectilinear room with a few partition
    walls, one ray per degree from -90 to +90 degrees, Gaussian range noise.  Rays that hit nothing within
    max_range_mm report max_range_mm (beyond the detector's 8 m limit).  Returns (range, lx, ly) float64 arrays."""
    rng = np.random.default_rng(seed)
    w, h = rng.uniform(3000.0, 9000.0, 2)          # room half-extents, mm
    rot = rng.uniform(-math.pi, math.pi)            # room orientation relative to the robot
    walls = [((-w, -h), (w, -h)), ((w, -h), (w, h)), ((w, h), (-w, h)), ((-w, h), (-w, -h))]
    for _ in range(int(rng.integers(0, 4))):       # partitions jutting out of a wall: more corners
        if rng.random() < 0.5:
            x0 = rng.uniform(-0.8 * w, 0.8 * w)
            walls.append(((x0, -h), (x0, -h + rng.uniform(0.2, 0.7) * 2 * h)))
        else:
            y0 = rng.uniform(-0.8 * h, 0.8 * h)
            walls.append(((-w, y0), (-w + rng.uniform(0.2, 0.7) * 2 * w, y0)))
    off = np.array([rng.uniform(-0.6 * w, 0.6 * w), rng.uniform(-0.6 * h, 0.6 * h)])
    c, s = math.cos(rot), math.sin(rot)
    Rm = np.array([[c, -s], [s, c]])
    segs = [(Rm @ (np.array(a) - off), Rm @ (np.array(b) - off)) for a, b in walls]
    ang = np.deg2rad(np.linspace(-90.0, 90.0, n_rays))
    rngs = np.full(n_rays, max_range_mm)
    for a, b in segs:
        d = b - a
        for k, th in enumerate(ang):
            u = np.array([math.cos(th), math.sin(th)])
            den = u[0] * d[1] - u[1] * d[0]
            if abs(den) < 1e-12:
                continue
            t = (a[0] * d[1] - a[1] * d[0]) / den      # distance along the ray
            q = (a[0] * u[1] - a[1] * u[0]) / den      # position along the wall
            if t > 50.0 and 0.0 <= q <= 1.0 and t < rngs[k]:
                rngs[k] = t
    hit = rngs < max_range_mm
    rngs = np.where(hit, np.maximum(rngs + noise_mm * rng.standard_normal(n_rays), 20.0), max_range_mm)
    rngs = np.floor(rngs)                              # ArSensorReading::getRange is an unsigned int of millimetres
    return rngs.astype(np.float64), rngs * np.cos(ang), rngs * np.sin(ang)


def _cast_rays(segs, ang, max_range_mm):
    """Distances along the rays (angles `ang`, from the origin) to the nearest of the wall segments (robot frame, mm)."""
    rngs = np.full(ang.size, max_range_mm)
    for a, b in segs:
        d = b - a
        ux, uy = np.cos(ang), np.sin(ang)
        den = ux * d[1] - uy * d[0]
        ok = np.abs(den) > 1e-12
        den = np.where(ok, den, 1.0)
        t = (a[0] * d[1] - a[1] * d[0]) / den
        q = (a[0] * uy - a[1] * ux) / den
        hit = ok & (t > 50.0) & (q >= 0.0) & (q <= 1.0) & (t < rngs)
        rngs = np.where(hit, t, rngs)
    return rngs


def simulated_drive(seed=20260012, steps=120, dt=0.25, v=0.3, noise_mm=6.0, max_range_mm=12000.0):
    """A robot driving a gentle arc through a 12 m x 9 m room with two partitions, one SICK sweep per loop iteration
    (181 rays, -90..+90 degrees) cast from the TRUE pose, odometry with the reference's noise model
    (kalmanfilter.cpp:28-37): the input of the whole slam.cpp loop (:130-204), perception included.
    Returns a list of dict(dt, v_mm_s, rot_deg_s, scan=(range_mm, lx, ly), truth)."""
    rng = np.random.default_rng(seed)
    W, H = 6000.0, 4500.0
    walls = [((-W, -H), (W, -H)), ((W, -H), (W, H)), ((W, H), (-W, H)), ((-W, H), (-W, -H)),
             ((1500.0, -H), (1500.0, -1200.0)), ((-W, 1000.0), (-2500.0, 1000.0))]
    walls = [(np.array(a), np.array(b)) for a, b in walls]
    pose = np.array([-3500.0, -2500.0, 0.35])  # mm, mm, rad in the room frame; the filter's frame starts at this pose
    w = 0.06
    ang = np.deg2rad(np.linspace(-90.0, 90.0, 181))
    out = []
    for s in range(steps):
        pose = pose + dt * np.array([1000.0 * v * math.cos(pose[2]), 1000.0 * v * math.sin(pose[2]), w])
        c, sn = math.cos(pose[2]), math.sin(pose[2])
        Rt = np.array([[c, sn], [-sn, c]])
        segs = [(Rt @ (a - pose[:2]), Rt @ (b - pose[:2])) for a, b in walls]
        r = _cast_rays(segs, ang, max_range_mm)
        hit = r < max_range_mm
        r = np.where(hit, np.maximum(r + noise_mm * rng.standard_normal(ang.size), 20.0), max_range_mm)
        r = np.floor(r)
        v_meas = v + SIGMA_V * v * rng.standard_normal()
        w_meas = w + SIGMA_W * v * rng.standard_normal()
        out.append(dict(dt=dt, v_mm_s=1000.0 * v_meas, rot_deg_s=w_meas * 180.0 / 3.141592654,
                        scan=(r.astype(np.float64), r * np.cos(ang), r * np.sin(ang)), truth=pose.copy()))
    return out


def state_digest(x, P):
    """sha256 over the state rounded to 9 significant digits: what two implementations within the parity tolerance normally
    share (a value that sits on a rounding boundary can still split them: the digest is a fingerprint, the tolerance is the test)."""
    import hashlib

    def rnd(a):
        a = np.asarray(a, dtype=np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            mag = np.where(a == 0, 0.0, np.floor(np.log10(np.abs(a))))
        return np.where(a == 0, 0.0, np.round(a / 10.0 ** mag, 8) * 10.0 ** mag)

    return hashlib.sha256(np.ascontiguousarray(rnd(x)).tobytes() + np.ascontiguousarray(rnd(P)).tobytes()).hexdigest()[:16]
