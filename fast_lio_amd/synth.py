"""Seeded synthetic scenes, scans and priors for the measurement-update hot path.

"Box-city": a ground plane z=0 over a square plus axis-aligned vertical wall slabs, sampled on a
0.5 m lattice (one point per filter_size_map voxel, the density ikd-Tree's down-sampling produces).
Scans are generated *post-downsample* (at most one point per filter_size_surf voxel), so N is exact;
this bypasses the reference's VoxelGrid (src/laserMapping.cpp:904-905) and Preprocess stages.

Pure numpy; no dependency on the oracle or on the GPU library.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

NDOF = 23
NSTATE = 26
G_LEN = 98090.0 / 10000.0  # S2<double,98090,10000,1>, include/use-ikfom.hpp:8

# flat state offsets
X_POS, X_ROT, X_OFFR, X_OFFT, X_VEL, X_BG, X_BA, X_GRAV = 0, 3, 7, 11, 14, 17, 20, 23


@dataclass
class Sensor:
    """FoV / range limits and LiDAR->IMU extrinsics (config/*.yaml of the reference)."""

    name: str
    az_deg: tuple  # (min, max) azimuth in the LiDAR frame, +x forward
    el_deg: tuple  # (min, max) elevation
    blind: float
    det_range: float
    extrinsic_T: tuple
    extrinsic_R: tuple = (1, 0, 0, 0, 1, 0, 0, 0, 1)
    height: float = 2.0  # sensor height above ground used by the default true pose
    voxel: float = 0.5   # leaf of the scan down-sampling the generated scan stands for (filter_size_surf)


SENSORS = {
    # config/avia.yaml:11,19-24 ; Avia FoV 70.4 x 77.2 deg
    # height 60 m: a UAV-borne Avia looking over the 20 m wall slabs -- a ground-level sensor in box-city
    # sees < 45k distinct 0.5 m voxels, short of the 100k-point stress scan BASELINE.json asks for
    "avia": Sensor("avia", (-35.2, 35.2), (-38.6, 38.6), 4.0, 450.0, (0.04165, 0.02326, -0.0284), height=60.0),
    # config/velodyne.yaml (extrinsic_T [0,0,0.28], det_range 100, blind 2), spinning 360 x +-15 deg
    "velodyne": Sensor("velodyne", (-180.0, 180.0), (-15.0, 15.0), 2.0, 100.0, (0.0, 0.0, 0.28), height=2.0),
    # config/ouster64.yaml (det_range 150, blind 4), 360 x +-22.5 deg
    "ouster64": Sensor("ouster64", (-180.0, 180.0), (-22.5, 22.5), 4.0, 150.0, (0.0, 0.0, 0.0), height=2.5),
    # config/mid360.yaml (extrinsic_T [-0.011,-0.02329,0.04412], det_range 100, blind 0.5), 360 x 59 deg
    # BASELINE configs[4] asks for DENSE 200k-point MID-360 scans: within det_range 100 m the scene shows < 100k distinct
    # 0.5 m voxels, so the dense scan is the one a 0.25 m leaf leaves (filter_size_surf 0.25)
    "mid360": Sensor("mid360", (-180.0, 180.0), (-7.0, 52.0), 0.5, 100.0, (-0.011, -0.02329, 0.04412), height=1.5, voxel=0.25),
}


@dataclass
class Scene:
    L: float                      # ground square side (m)
    walls: np.ndarray             # W x 5: axis(0=x-aligned wall in plane y=c, 1=plane x=c), c, lo, hi, height
    seed: int
    map_xyz: np.ndarray = field(repr=False, default=None)  # M x 3 float32


# ------------------------------------------------------------------ quaternion helpers (xyzw)
def quat_from_rotvec(v):
    v = np.asarray(v, np.float64)
    th = np.linalg.norm(v)
    if th < 1e-300:
        return np.array([0.0, 0.0, 0.0, 1.0])
    s = np.sin(th / 2) / th
    return np.array([v[0] * s, v[1] * s, v[2] * s, np.cos(th / 2)])


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx,
        aw * bw - ax * bx - ay * by - az * bz,
    ])


def quat_to_R(q):
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def R_to_quat(R):
    R = np.asarray(R, np.float64).reshape(3, 3)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    return q / np.linalg.norm(q)


def make_state(pos=(0, 0, 0), rot=(0, 0, 0, 1), offR=(0, 0, 0, 1), offT=(0, 0, 0), vel=(0, 0, 0), bg=(0, 0, 0),
               ba=(0, 0, 0), grav=(0, 0, -G_LEN)):
    x = np.zeros(NSTATE)
    x[X_POS:X_POS + 3] = pos
    x[X_ROT:X_ROT + 4] = rot
    x[X_OFFR:X_OFFR + 4] = offR
    x[X_OFFT:X_OFFT + 3] = offT
    x[X_VEL:X_VEL + 3] = vel
    x[X_BG:X_BG + 3] = bg
    x[X_BA:X_BA + 3] = ba
    x[X_GRAV:X_GRAV + 3] = grav
    return x


# ------------------------------------------------------------------ map
def make_scene(M: int, seed: int, wall_len: float = 100.0, wall_h: float = 20.0, spacing: float = 0.5) -> Scene:
    """Box-city with exactly M map points: ~80 % ground lattice, rest on wall slabs."""
    rng = np.random.default_rng(seed)
    side = max(2, int(round(np.sqrt(0.8 * M))))
    while side * side > M:
        side -= 1
    L = side * spacing
    n_g = side * side
    rem = M - n_g
    cols = int(round(wall_len / spacing))
    rows = int(round(wall_h / spacing))
    per_wall = cols * rows
    n_walls = int(np.ceil(rem / per_wall)) if rem > 0 else 0
    walls = np.zeros((n_walls, 5))
    half = L / 2
    for w in range(n_walls):
        for _ in range(1000):
            axis = int(rng.integers(0, 2))
            wl = min(wall_len, L * 0.8)
            c = (np.floor(rng.uniform(-half + 1, half - 1) / spacing) + 0.5) * spacing
            lo = np.floor(rng.uniform(-half, half - wl) / spacing) * spacing
            hi = lo + wl
            # keep the sensor neighbourhood (origin) free: wall plane >= 8 m away or span not covering it
            if abs(c) < 8.0 and lo < 8.0 and hi > -8.0:
                continue
            walls[w] = (axis, c, lo, hi, wall_h)
            break
        else:
            walls[w] = (0, half - 1.0, -half * 0.8, -half * 0.8 + min(wall_len, L * 0.8), wall_h)
    # ground lattice
    gi = (np.arange(side) + 0.5) * spacing - half
    gx, gy = np.meshgrid(gi, gi, indexing="ij")
    g = np.empty((n_g, 3), np.float64)
    g[:, 0] = gx.ravel() + rng.uniform(-0.1, 0.1, n_g)
    g[:, 1] = gy.ravel() + rng.uniform(-0.1, 0.1, n_g)
    g[:, 2] = rng.normal(0.0, 0.01, n_g)
    parts = [g]
    left = rem
    for w in range(n_walls):
        axis, c, lo, hi, H = walls[w]
        wl = hi - lo
        nc = int(round(wl / spacing))
        nr = rows
        u = (np.arange(nc) + 0.5) * spacing + lo
        v = (np.arange(nr) + 0.5) * spacing
        uu, vv = np.meshgrid(u, v, indexing="ij")
        n = uu.size
        pts = np.empty((n, 3))
        along = uu.ravel() + rng.uniform(-0.1, 0.1, n)
        up = vv.ravel() + rng.uniform(-0.1, 0.1, n)
        off = c + rng.normal(0.0, 0.01, n)
        if int(axis) == 0:  # wall in plane y = c, extends along x
            pts[:, 0], pts[:, 1], pts[:, 2] = along, off, up
        else:              # wall in plane x = c, extends along y
            pts[:, 0], pts[:, 1], pts[:, 2] = off, along, up
        take = min(left, n)
        parts.append(pts[:take])
        left -= take
    xyz = np.concatenate(parts, axis=0)
    if xyz.shape[0] < M:  # tiny M corner case: pad with extra jittered ground points
        extra = M - xyz.shape[0]
        e = np.empty((extra, 3))
        e[:, 0] = rng.uniform(-half, half, extra)
        e[:, 1] = rng.uniform(-half, half, extra)
        e[:, 2] = rng.normal(0.0, 0.01, extra)
        xyz = np.concatenate([xyz, e], axis=0)
    xyz = xyz[:M]
    perm = rng.permutation(M)  # map order is arbitrary in the reference (insertion order)
    sc = Scene(L=L, walls=walls, seed=seed)
    sc.map_xyz = np.ascontiguousarray(xyz[perm], dtype=np.float32)
    return sc


# ------------------------------------------------------------------ scan
def _raycast(scene: Scene, o: np.ndarray, d: np.ndarray, rmin: float, rmax: float) -> np.ndarray:
    """Nearest hit range for rays o + t d (d unit) against ground + walls; inf where no hit."""
    n = d.shape[0]
    best = np.full(n, np.inf)
    half = scene.L / 2
    with np.errstate(divide="ignore", invalid="ignore"):
        t = -o[2] / d[:, 2]
        x = o[0] + t * d[:, 0]
        y = o[1] + t * d[:, 1]
        ok = (t > 0) & (np.abs(x) <= half) & (np.abs(y) <= half)
        best = np.where(ok, t, best)
        for axis, c, lo, hi, H in scene.walls:
            a = int(axis)
            nrm, alo = (1, 0) if a == 0 else (0, 1)
            t = (c - o[nrm]) / d[:, nrm]
            al = o[alo] + t * d[:, alo]
            z = o[2] + t * d[:, 2]
            ok = (t > 0) & (al >= lo) & (al <= hi) & (z >= 0) & (z <= H) & (t < best)
            best = np.where(ok, t, best)
    best = np.where((best >= rmin) & (best <= rmax), best, np.inf)
    return best


def make_scan(scene: Scene, sensor: Sensor, N: int, x_true: np.ndarray, seed: int, voxel: float | None = None,
              range_noise: float = 0.02) -> np.ndarray:
    """N post-downsample LiDAR-frame points (float32 N x 3) seen from the true state x_true."""
    if voxel is None:
        voxel = sensor.voxel
    rng = np.random.default_rng(seed)
    R = quat_to_R(x_true[X_ROT:X_ROT + 4])
    R_LI = quat_to_R(x_true[X_OFFR:X_OFFR + 4])
    t_LI = x_true[X_OFFT:X_OFFT + 3]
    t = x_true[X_POS:X_POS + 3]
    o = R @ t_LI + t
    Rw = R @ R_LI
    all_pts = np.zeros((0, 3))
    all_keys = np.zeros((0,), np.int64)
    batch = max(4 * N, 20000)
    stalled = 0
    for _ in range(64):
        az = np.deg2rad(rng.uniform(sensor.az_deg[0], sensor.az_deg[1], batch))
        # 40 % of the rays uniform in elevation (walls, near field); 60 % aimed so that their ground
        # footprint is uniform in AREA out to det_range -- otherwise the far field, where one 0.5 m voxel
        # subtends ~0.01 deg of elevation, never fills up and N distinct voxels are unreachable
        u = rng.uniform(0, 1, batch)
        el_lo, el_hi = np.deg2rad(sensor.el_deg[0]), np.deg2rad(sensor.el_deg[1])
        h_eff = max(float(o[2]), 0.2)
        rg = np.sqrt(rng.uniform(sensor.blind ** 2, min(sensor.det_range, 0.75 * scene.L) ** 2, batch))
        el_ground = np.clip(-np.arctan2(h_eff, rg), el_lo, el_hi)
        el = np.where(rng.uniform(0, 1, batch) < 0.4, el_lo + (el_hi - el_lo) * u, el_ground)
        dl = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=1)
        dw = dl @ Rw.T
        r = _raycast(scene, o, dw, sensor.blind, sensor.det_range)
        ok = np.isfinite(r)
        r = r[ok] + rng.normal(0.0, range_noise, int(ok.sum()))
        pl = dl[ok] * r[:, None]
        # voxel de-dup in the LiDAR frame (the reference down-samples feats_undistort, body frame):
        # keep the first sample that lands in each voxel
        key = np.floor(pl / voxel).astype(np.int64) + (1 << 20)
        k1 = (key[:, 0] << 42) | (key[:, 1] << 21) | key[:, 2]
        all_keys_before = all_keys.shape[0]
        all_pts = np.concatenate([all_pts, pl], axis=0)
        all_keys = np.concatenate([all_keys, k1], axis=0)
        _, first = np.unique(all_keys, return_index=True)
        first.sort()
        grew = len(first) - all_keys_before
        all_pts = all_pts[first]
        all_keys = all_keys[first]
        if all_pts.shape[0] >= N:
            break
        stalled = stalled + 1 if grew < max(N // 500, 1) else 0
        if stalled >= 3:  # the visible surface is exhausted: more rays will not find new voxels
            break
    pts = all_pts
    if pts.shape[0] < N:
        raise RuntimeError(f"scene too small for a {N}-point scan (got {pts.shape[0]} voxels)")
    sel = rng.permutation(pts.shape[0])[:N]
    return np.ascontiguousarray(pts[np.sort(sel)], dtype=np.float32)


# ------------------------------------------------------------------ truth / prior
def _probe_coverage(scene: Scene, sensor: Sensor, x: np.ndarray, rng, n_rays: int = 40000, voxel: float = 0.5) -> int:
    R = quat_to_R(x[X_ROT:X_ROT + 4])
    Rw = R @ quat_to_R(x[X_OFFR:X_OFFR + 4])
    o = R @ x[X_OFFT:X_OFFT + 3] + x[X_POS:X_POS + 3]
    az = np.deg2rad(rng.uniform(sensor.az_deg[0], sensor.az_deg[1], n_rays))
    el = np.deg2rad(rng.uniform(sensor.el_deg[0], sensor.el_deg[1], n_rays))
    dl = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=1)
    r = _raycast(scene, o, dl @ Rw.T, sensor.blind, sensor.det_range)
    ok = np.isfinite(r)
    key = np.floor(dl[ok] * r[ok][:, None] / voxel).astype(np.int64) + (1 << 20)
    return int(np.unique((key[:, 0] << 42) | (key[:, 1] << 21) | key[:, 2]).size)


def true_state(sensor: Sensor, seed: int, scene: Scene | None = None) -> np.ndarray:
    """A plausible true state near the scene centre: small roll/pitch, sensor height, and (when a
    scene is given) the yaw among 12 seeded candidates that sees the most distinct voxels, so a
    narrow-FoV sensor is not parked facing a wall."""
    rng = np.random.default_rng(seed ^ 0x5EED)
    yaw0 = rng.uniform(-np.pi, np.pi)
    rp = np.deg2rad(rng.uniform(-2.0, 2.0, 2))
    # small test scenes cannot be seen from 60 m up (the nearest ground return would fall outside the
    # square): scale the height with the scene, 0.08 * L capped by the sensor's nominal height
    hgt = sensor.height if scene is None else max(2.0, min(sensor.height, 0.08 * scene.L))
    pos = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), hgt])
    vel = rng.normal(0, 0.2, 3)
    bg = rng.normal(0, 1e-3, 3)
    ba = rng.normal(0, 1e-2, 3)
    offR = R_to_quat(np.asarray(sensor.extrinsic_R, float).reshape(3, 3))

    def build(yaw):
        q = quat_mul(quat_from_rotvec([0, 0, yaw]), quat_mul(quat_from_rotvec([0, rp[0], 0]), quat_from_rotvec([rp[1], 0, 0])))
        return make_state(pos=pos, rot=q, offR=offR, offT=sensor.extrinsic_T, vel=vel, bg=bg, ba=ba)

    if scene is None or (sensor.az_deg[1] - sensor.az_deg[0]) >= 359.0:
        return build(yaw0)
    best, best_cov = None, -1
    for k in range(12):
        x = build(yaw0 + k * (2 * np.pi / 12))
        cov = _probe_coverage(scene, sensor, x, np.random.default_rng(seed + 17))
        if cov > best_cov:
            best, best_cov = x, cov
    return best


def perturb_prior(x_true: np.ndarray, seed: int, pos_err: float = 0.05, rot_err_deg: float = 0.5) -> np.ndarray:
    """x_prior = truth [+] delta with pos U(-pos_err, pos_err), rot U(-rot_err, rot_err)."""
    rng = np.random.default_rng(seed ^ 0xBEEF)
    x = x_true.copy()
    x[X_POS:X_POS + 3] += rng.uniform(-pos_err, pos_err, 3)
    dth = np.deg2rad(rng.uniform(-rot_err_deg, rot_err_deg, 3))
    x[X_ROT:X_ROT + 4] = quat_mul(x[X_ROT:X_ROT + 4], quat_from_rotvec(dth))
    return x


def init_P() -> np.ndarray:
    """IMU_init covariance (src/IMU_Processing.hpp:204-210)."""
    P = np.eye(NDOF)
    P[6:9, 6:9] = np.eye(3) * 0.00001
    P[9:12, 9:12] = np.eye(3) * 0.00001
    P[15:18, 15:18] = np.eye(3) * 0.0001
    P[18:21, 18:21] = np.eye(3) * 0.001
    P[21:23, 21:23] = np.eye(2) * 0.00001
    return P


def process_noise_cov() -> np.ndarray:
    """use-ikfom.hpp:35-43."""
    Q = np.zeros((12, 12))
    Q[0:3, 0:3] = np.eye(3) * 0.0001
    Q[3:6, 3:6] = np.eye(3) * 0.0001
    Q[6:9, 6:9] = np.eye(3) * 0.00001
    Q[9:12, 9:12] = np.eye(3) * 0.00001
    return Q


def propagate_prior_cov(predict_fn, x_prior: np.ndarray, n_steps: int = 10, rate_hz: float = 200.0):
    """P^- = IMU_init diagonal pushed through n_steps predict() calls (SURVEY 8d "Prior").

    predict_fn(x, P, dt, Q, acc, gyro) -> (x, P) is the caller's esekf::predict (product host library in
    bench/smoke, oracle in the oracle-only tests).  The state is NOT advanced (we keep x_prior): the
    IMU sample is chosen so the platform is static (acc cancels gravity, gyro = bias).
    """
    P = init_P()
    Q = process_noise_cov()
    dt = 1.0 / rate_hz
    x = x_prior.copy()
    R = quat_to_R(x[X_ROT:X_ROT + 4])
    acc = R.T @ (-x[X_GRAV:X_GRAV + 3]) + x[X_BA:X_BA + 3]
    gyro = x[X_BG:X_BG + 3].copy()
    vel0 = x[X_VEL:X_VEL + 3].copy()
    for _ in range(n_steps):
        x[X_VEL:X_VEL + 3] = 0.0
        x, P = predict_fn(x, P, dt, Q, acc, gyro)
    xp = x_prior.copy()
    xp[X_VEL:X_VEL + 3] = vel0
    return xp, P


@dataclass
class Problem:
    scene: Scene
    sensor: Sensor
    x_true: np.ndarray
    x_prior: np.ndarray
    body: np.ndarray  # N x 3 float32

    @property
    def map_xyz(self):
        return self.scene.map_xyz


CONFIG_SEED_BASE = 0xFA570000


def make_problem(M: int, N: int, sensor: str = "avia", cfg: int = 0, scan_seed: int = 0, scene: Scene | None = None) -> Problem:
    """Scene + one scan + perturbed prior state.  seed = 0xFA570000 + cfg."""
    seed = CONFIG_SEED_BASE + cfg
    sn = SENSORS[sensor]
    if scene is None:
        scene = make_scene(M, seed)
    xt = true_state(sn, seed + 1000 * scan_seed, scene)
    body = make_scan(scene, sn, N, xt, seed + 7 + 1000 * scan_seed)
    xp = perturb_prior(xt, seed + 13 + 1000 * scan_seed)
    return Problem(scene=scene, sensor=sn, x_true=xt, x_prior=xp, body=body)


def imu_poses(x0, predict_fn, n_imu: int = 21, T: float = 0.1, seed: int = 4):
    """IMUpose + scan-end state as UndistortPcl's forward half builds them (IMU_Processing.hpp:240-300): one predict
    per synthetic IMU sample.  Returns (ctypes array of 22-double Pose6D records, x_end)."""
    import ctypes as C

    class Pose6D(C.Structure):
        _fields_ = [("offset_time", C.c_double), ("acc", C.c_double * 3), ("gyr", C.c_double * 3), ("vel", C.c_double * 3),
                    ("pos", C.c_double * 3), ("rot", C.c_double * 9)]

    def rotm(q):
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    rng = np.random.default_rng(seed)
    x = np.array(x0, np.float64)
    x[14:17] = (8.0, -3.0, 0.5)
    P = init_P()
    Q = process_noise_cov()
    dt = T / (n_imu - 1)
    arr = (Pose6D * n_imu)()

    def put(k, t, a, g, xs):
        arr[k].offset_time = float(t)
        R = rotm(xs[3:7]).reshape(9)
        for i in range(3):
            arr[k].acc[i], arr[k].gyr[i], arr[k].vel[i], arr[k].pos[i] = float(a[i]), float(g[i]), float(xs[14 + i]), float(xs[i])
        for i in range(9):
            arr[k].rot[i] = float(R[i])

    put(0, 0.0, (0, 0, 0), (0, 0, 0), x)
    for k in range(1, n_imu):
        gyro = np.array([0.4, -0.3, 0.9]) + rng.normal(0, 0.05, 3)
        acc = np.array([0.8, -0.5, 9.9]) + rng.normal(0, 0.2, 3)
        x, P = predict_fn(x, P, dt, Q, acc, gyro)
        acc_s = rotm(x[3:7]) @ (acc - x[20:23]) + x[23:26]
        put(k, k * dt, acc_s, gyro - x[17:20], x)
    return arr, x
