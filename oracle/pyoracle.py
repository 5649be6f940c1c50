"""ctypes binding of the CPU oracle (oracle/libfastlio_oracle.so).

TEST INFRASTRUCTURE ONLY -- see oracle/fastlio_oracle.h.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product (fast_lio_amd/) never does.
PARITY UNPINNED: the reference ships no golden vectors and cannot be built here.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfastlio_oracle.so")
NDOF = 23
NSTATE = 26
K = 5


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (idempotent)."""
    srcs = [os.path.join(_HERE, f) for f in ("oracle_math.c", "oracle_path.c", "fastlio_oracle.h", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB_PATH


_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


class _Scan(C.Structure):
    _fields_ = [
        ("N", C.c_int),
        ("body", C.POINTER(C.c_float)),
        ("world", C.POINTER(C.c_float)),
        ("nn_idx", C.POINTER(C.c_int32)),
        ("nn_d2", C.POINTER(C.c_float)),
        ("nn_cnt", C.POINTER(C.c_uint8)),
        ("selected", C.POINTER(C.c_uint8)),
        ("normvec", C.POINTER(C.c_float)),
        ("res_last", C.POINTER(C.c_float)),
        ("effct_feat_num", C.c_int),
        ("total_residual", C.c_double),
        ("res_mean_last", C.c_double),
        ("h_x", C.POINTER(C.c_double)),
        ("h", C.POINTER(C.c_double)),
        ("cap_rows", C.c_int),
        ("match_time", C.c_double),
        ("solve_time", C.c_double),
        ("nthreads", C.c_int),
        ("search_radius2", C.c_double),
    ]


class UpdateStats(C.Structure):
    _fields_ = [
        ("passes", C.c_int),
        ("searches", C.c_int),
        ("returned_in_loop", C.c_int),
        ("n_eff", C.c_int * 8),
        ("pass_search", C.c_int * 8),
        ("h_time", C.c_double),
        ("solve_time", C.c_double),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    L.orc_kdtree_build.restype = C.c_void_p
    L.orc_kdtree_build.argtypes = [_f32p, C.c_size_t, C.c_size_t]
    L.orc_kdtree_free.argtypes = [C.c_void_p]
    L.orc_knn5.restype = C.c_int
    L.orc_knn5.argtypes = [C.c_void_p, _f32p, _i32p, _f32p]
    L.orc_knn5_brute.restype = C.c_int
    L.orc_knn5_brute.argtypes = [_f32p, C.c_size_t, C.c_size_t, _f32p, _i32p, _f32p]
    L.orc_knn5_batch.argtypes = [C.c_void_p, _f32p, C.c_size_t, _i32p, _f32p, _u8p, C.c_int]
    L.orc_esti_plane.restype = C.c_int
    L.orc_esti_plane.argtypes = [_f32p, C.c_float, _f32p]
    L.orc_qr_solve_5x3.argtypes = [_f32p, _f32p, _f32p]
    L.orc_points_body_to_world.argtypes = [_f64p, _f32p, C.c_size_t, C.c_size_t, _f32p]
    L.orc_set_eigen_order.argtypes = [C.c_int]
    L.orc_get_eigen_order.restype = C.c_int
    L.orc_A_matrix.argtypes = [_f64p, _f64p]
    L.orc_so3_exp.argtypes = [_f64p, C.c_double, _f64p]
    L.orc_so3_log.argtypes = [_f64p, _f64p]
    L.orc_quat_mul.argtypes = [_f64p, _f64p, _f64p]
    L.orc_quat_rot.argtypes = [_f64p, _f64p, _f64p]
    L.orc_S2_Bx.argtypes = [_f64p, _f64p]
    L.orc_S2_Nx_yy.argtypes = [_f64p, _f64p]
    L.orc_S2_Mx.argtypes = [_f64p, _f64p, _f64p]
    L.orc_S2_boxplus.argtypes = [_f64p, _f64p]
    L.orc_S2_boxminus.argtypes = [_f64p, _f64p, _f64p]
    L.orc_state_boxplus.argtypes = [_f64p, _f64p]
    L.orc_state_boxminus.argtypes = [_f64p, _f64p, _f64p]
    L.orc_inverse.restype = C.c_int
    L.orc_inverse.argtypes = [_f64p, C.c_int, _f64p]
    L.orc_predict.argtypes = [_f64p, _f64p, C.c_double, _f64p, _f64p, _f64p]
    L.orc_process_noise_cov.argtypes = [_f64p]
    L.orc_init_P.argtypes = [_f64p]
    L.orc_scan_create.restype = C.POINTER(_Scan)
    L.orc_scan_create.argtypes = [_f32p, C.c_size_t, C.c_int]
    L.orc_scan_free.argtypes = [C.POINTER(_Scan)]
    L.orc_scan_reset.argtypes = [C.POINTER(_Scan)]
    L.orc_h_share_model.restype = C.c_int
    L.orc_h_share_model.argtypes = [C.POINTER(_Scan), C.c_void_p, _f32p, C.c_size_t, _f64p, C.c_int, C.c_int]
    L.orc_normal_equations.argtypes = [C.POINTER(_Scan), _f64p, _f64p]
    L.orc_update_iterated.argtypes = [
        C.POINTER(_Scan), C.c_void_p, _f32p, C.c_size_t, _f64p, _f64p, C.c_double, C.c_int, _f64p, C.c_int,
        C.POINTER(UpdateStats),
    ]
    L.orc_iekf_pass_info.argtypes = [_f64p, _f64p, _f64p, C.c_double, _f64p, _f64p, _f64p, _f64p, _f64p]
    L.orc_iekf_pass_gain.argtypes = [_f64p, _f64p, _f64p, C.c_double, _f64p, _f64p, C.c_int, _f64p, _f64p, _f64p]
    L.orc_map_incremental_classify.argtypes = [
        C.POINTER(_Scan), _f32p, C.c_size_t, _f64p, C.c_double, C.c_int, _f32p, _u8p,
    ]
    L.orc_map_add.restype = C.c_size_t
    L.orc_map_add.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_int, C.c_double]
    L.orc_map_add_floatbox.restype = C.c_size_t
    L.orc_map_add_floatbox.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_float]
    L.orc_map_delete_boxes.restype = C.c_size_t
    L.orc_map_delete_boxes.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t]
    _lib = L
    return L


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Map:
    """Static map + exact 5-NN k-d tree (ikd-Tree stand-in)."""

    def __init__(self, xyz: np.ndarray):
        self.xyz = _c32(xyz).reshape(-1, 3)
        self.M = self.xyz.shape[0]
        self._t = lib().orc_kdtree_build(self.xyz, 3, self.M)

    def __del__(self):
        try:
            if self._t:
                lib().orc_kdtree_free(self._t)
                self._t = None
        except Exception:
            pass

    def knn5(self, q):
        idx = np.empty(K, np.int32)
        d2 = np.empty(K, np.float32)
        n = lib().orc_knn5(self._t, _c32(q), idx, d2)
        return n, idx, d2

    def knn5_batch(self, q, nthreads=0):
        q = _c32(q).reshape(-1, 3)
        n = q.shape[0]
        idx = np.empty((n, K), np.int32)
        d2 = np.empty((n, K), np.float32)
        cnt = np.empty(n, np.uint8)
        lib().orc_knn5_batch(self._t, q, n, idx, d2, cnt, nthreads)
        return idx, d2, cnt

    def knn5_brute(self, q):
        idx = np.empty(K, np.int32)
        d2 = np.empty(K, np.float32)
        n = lib().orc_knn5_brute(self.xyz, 3, self.M, _c32(q), idx, d2)
        return n, idx, d2


class Scan:
    """The per-scan globals of h_share_model (laserMapping.cpp:76-114)."""

    def __init__(self, body_xyz: np.ndarray, nthreads: int = 3):
        b = _c32(body_xyz).reshape(-1, 3)
        self.N = b.shape[0]
        self._s = lib().orc_scan_create(b, 3, self.N)
        self._s.contents.nthreads = nthreads

    def __del__(self):
        try:
            if self._s:
                lib().orc_scan_free(self._s)
                self._s = None
        except Exception:
            pass

    def reset(self):
        lib().orc_scan_reset(self._s)

    @property
    def c(self):
        return self._s.contents

    def set_threads(self, n):
        self._s.contents.nthreads = n

    def set_search_radius2(self, r2):
        self._s.contents.search_radius2 = r2

    def _arr(self, ptr, shape, dtype):
        n = int(np.prod(shape))
        if n == 0:
            return np.zeros(shape, dtype)
        return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).reshape(shape).copy()

    @property
    def selected(self):
        return self._arr(self.c.selected, (self.N,), np.uint8)

    @property
    def nn_idx(self):
        return self._arr(self.c.nn_idx, (self.N, K), np.int32)

    @property
    def nn_d2(self):
        return self._arr(self.c.nn_d2, (self.N, K), np.float32)

    @property
    def nn_cnt(self):
        return self._arr(self.c.nn_cnt, (self.N,), np.uint8)

    @property
    def world(self):
        return self._arr(self.c.world, (self.N, 3), np.float32)

    @property
    def normvec(self):
        return self._arr(self.c.normvec, (self.N, 4), np.float32)

    @property
    def n_eff(self):
        return self.c.effct_feat_num

    @property
    def total_residual(self):
        return self.c.total_residual

    @property
    def h_x(self):
        """n_eff x 12 (converted from the column-major storage)."""
        n = self.n_eff
        return self._arr(self.c.h_x, (12, n), np.float64).T.copy()

    @property
    def h(self):
        return self._arr(self.c.h, (self.n_eff,), np.float64)

    def h_share_model(self, m: Map, x, converge=True, extrinsic_est_en=False) -> bool:
        return bool(lib().orc_h_share_model(self._s, m._t, m.xyz, 3, _c64(x), int(converge), int(extrinsic_est_en)))

    def normal_equations(self):
        HTH = np.zeros((12, 12))
        HTh = np.zeros(12)
        lib().orc_normal_equations(self._s, HTH, HTh)
        return HTH, HTh

    def update_iterated(self, m: Map, x, P, R=0.001, max_iter=3, limit=None, extrinsic_est_en=False):
        x = _c64(x).copy()
        P = _c64(P).reshape(NDOF, NDOF).copy()
        limit = _c64(np.full(NDOF, 0.001) if limit is None else limit)
        st = UpdateStats()
        lib().orc_update_iterated(self._s, m._t, m.xyz, 3, x, P, R, max_iter, limit, int(extrinsic_est_en), C.byref(st))
        return x, P, st

    def map_incremental_classify(self, m: Map, x, filter_size_map=0.5, flg_EKF_inited=True):
        world = np.zeros((self.N, 3), np.float32)
        cls = np.zeros(self.N, np.uint8)
        lib().orc_map_incremental_classify(self._s, m.xyz, 3, _c64(x), filter_size_map, int(flg_EKF_inited), world, cls)
        return world, cls


# ---- thin functional wrappers for the KATs ----
def points_body_to_world(x, pts):
    """RGBpointBodyToWorld over a cloud (src/laserMapping.cpp:200-211, 478-530)."""
    a = _c32(pts)
    out = np.empty((a.shape[0], 3), np.float32)
    lib().orc_points_body_to_world(np.ascontiguousarray(x, np.float64), a.reshape(-1), a.shape[1], a.shape[0], out.reshape(-1))
    return out


ORDER_SEQ, ORDER_SSE, ORDER_PAIRWISE, ORDER_NOVEC = 0, 1, 2, 3
ORDER_NAMES = {0: "seq", 1: "sse", 2: "pairwise", 3: "novec"}


def set_eigen_order(order: int) -> None:
    """Process-wide fp32 summation order of the restated Eigen reductions (oracle_math.c "SUMMATION ORDER")."""
    lib().orc_set_eigen_order(int(order))


def get_eigen_order() -> int:
    return int(lib().orc_get_eigen_order())


def esti_plane(pts, threshold=0.1):
    out = np.zeros(4, np.float32)
    ok = lib().orc_esti_plane(_c32(pts).reshape(15), threshold, out)
    return bool(ok), out


def qr_solve_5x3(A, b):
    x = np.zeros(3, np.float32)
    lib().orc_qr_solve_5x3(_c32(A).reshape(15), _c32(b).reshape(5), x)
    return x


def A_matrix(v):
    A = np.zeros(9)
    lib().orc_A_matrix(_c64(v), A)
    return A.reshape(3, 3)


def so3_exp(v, scale=1.0):
    q = np.zeros(4)
    lib().orc_so3_exp(_c64(v), scale, q)
    return q


def so3_log(q):
    v = np.zeros(3)
    lib().orc_so3_log(_c64(q), v)
    return v


def quat_mul(a, b):
    o = np.zeros(4)
    lib().orc_quat_mul(_c64(a), _c64(b), o)
    return o


def quat_rot(q, v):
    o = np.zeros(3)
    lib().orc_quat_rot(_c64(q), _c64(v), o)
    return o


def S2_Bx(g):
    o = np.zeros(6)
    lib().orc_S2_Bx(_c64(g), o)
    return o.reshape(3, 2)


def S2_Nx_yy(g):
    o = np.zeros(6)
    lib().orc_S2_Nx_yy(_c64(g), o)
    return o.reshape(2, 3)


def S2_Mx(g, delta):
    o = np.zeros(6)
    lib().orc_S2_Mx(_c64(g), _c64(delta), o)
    return o.reshape(3, 2)


def S2_boxplus(g, delta):
    g = _c64(g).copy()
    lib().orc_S2_boxplus(g, _c64(delta))
    return g


def S2_boxminus(g, other):
    r = np.zeros(2)
    lib().orc_S2_boxminus(_c64(g), _c64(other), r)
    return r


def state_boxplus(x, dx):
    x = _c64(x).copy()
    lib().orc_state_boxplus(x, _c64(dx))
    return x


def state_boxminus(x, y):
    d = np.zeros(NDOF)
    lib().orc_state_boxminus(_c64(x), _c64(y), d)
    return d


def inverse(A):
    A = _c64(A)
    n = A.shape[0]
    o = np.zeros((n, n))
    lib().orc_inverse(A, n, o)
    return o


def init_P():
    P = np.zeros((NDOF, NDOF))
    lib().orc_init_P(P)
    return P


def process_noise_cov():
    Q = np.zeros((12, 12))
    lib().orc_process_noise_cov(Q)
    return Q


def predict(x, P, dt, Q, acc, gyro):
    x = _c64(x).copy()
    P = _c64(P).reshape(NDOF, NDOF).copy()
    lib().orc_predict(x, P, dt, _c64(Q), _c64(acc), _c64(gyro))
    return x, P


def iekf_pass_info(x, x_prop, P_prop, R, HTH, HTh):
    x = _c64(x).copy()
    P = np.zeros((NDOF, NDOF))
    Kx = np.zeros((NDOF, NDOF))
    dx = np.zeros(NDOF)
    lib().orc_iekf_pass_info(x, _c64(x_prop), _c64(P_prop), R, _c64(HTH), _c64(HTh), P, Kx, dx)
    return x, P, Kx, dx


def iekf_pass_gain(x, x_prop, P_prop, R, h_x, h):
    """h_x: n_eff x 12 row-major numpy -> passed column-major."""
    x = _c64(x).copy()
    h_x = _c64(h_x)
    n = h_x.shape[0]
    hcm = np.ascontiguousarray(h_x.T)
    P = np.zeros((NDOF, NDOF))
    Kx = np.zeros((NDOF, NDOF))
    dx = np.zeros(NDOF)
    lib().orc_iekf_pass_gain(x, _c64(x_prop), _c64(P_prop), R, hcm, _c64(h), n, P, Kx, dx)
    return x, P, Kx, dx


def map_add(map_xyz, add_xyz, downsample=True, ds=0.5):
    """ikd-Tree Add_Points stand-in: returns the new map array."""
    m = _c32(map_xyz).reshape(-1, 3)
    a = _c32(add_xyz).reshape(-1, 3)
    buf = np.zeros((len(m) + len(a), 3), np.float32)
    buf[: len(m)] = m
    n = lib().orc_map_add(buf, len(m), a, len(a), int(downsample), float(ds))
    return buf[:n].copy()


def map_add_floatbox(map_xyz, add_xyz, ds=0.5):
    """Sensitivity variant of map_add(downsample=True): ikd-Tree's float box arithmetic (oracle_path.c)."""
    m = _c32(map_xyz).reshape(-1, 3)
    a = _c32(add_xyz).reshape(-1, 3)
    buf = np.zeros((len(m) + len(a), 3), np.float32)
    buf[: len(m)] = m
    n = lib().orc_map_add_floatbox(buf, len(m), a, len(a), float(ds))
    return buf[:n].copy()


def map_delete_boxes(map_xyz, boxes):
    """ikd-Tree Delete_Point_Boxes stand-in: boxes (nb x 6: min xyz, max xyz)."""
    m = _c32(map_xyz).reshape(-1, 3).copy()
    b = _c32(boxes).reshape(-1, 6)
    n = lib().orc_map_delete_boxes(m, len(m), b, len(b))
    return m[:n].copy()


class LocalMap(C.Structure):
    """orc_local_map: LocalMap_Points + Localmap_Initialized."""
    _fields_ = [("vertex_min", C.c_float * 3), ("vertex_max", C.c_float * 3), ("initialized", C.c_int)]


def fov_segment(lm: LocalMap, pos_lid, cube_len=200.0, det_range=300.0):
    """lasermap_fov_segment restated: returns the slabs to delete (nb x 6)."""
    L = lib()
    L.orc_fov_segment.restype = C.c_int
    L.orc_fov_segment.argtypes = [C.POINTER(LocalMap), np.ctypeslib.ndpointer(np.float64), C.c_double, C.c_float,
                                  np.ctypeslib.ndpointer(np.float32)]
    boxes = np.zeros((3, 6), np.float32)
    nb = L.orc_fov_segment(C.byref(lm), _c64(pos_lid), float(cube_len), float(det_range), boxes)
    return boxes[:nb].copy()


def voxel_grid(xyz, leaf=0.5):
    """pcl::VoxelGrid restated (xyz centroids, ascending voxel index)."""
    a = _c32(xyz).reshape(-1, 3)
    L = lib()
    L.orc_voxel_grid.restype = C.c_size_t
    L.orc_voxel_grid.argtypes = [np.ctypeslib.ndpointer(np.float32), C.c_size_t, C.c_size_t, C.c_float,
                                 np.ctypeslib.ndpointer(np.float32)]
    out = np.zeros((max(len(a), 1), 3), np.float32)
    m = L.orc_voxel_grid(a, 3, len(a), float(leaf), out)
    return out[:m].copy()


class Pose6D(C.Structure):
    """msg/Pose6D.msg as the reference's IMUpose entries carry it."""
    _fields_ = [("offset_time", C.c_double), ("acc", C.c_double * 3), ("gyr", C.c_double * 3), ("vel", C.c_double * 3),
                ("pos", C.c_double * 3), ("rot", C.c_double * 9)]


def make_poses(rows):
    """rows: iterable of (offset_time, acc3, gyr3, vel3, pos3, rot9)."""
    arr = (Pose6D * len(rows))()
    for k, (t, a, g, v, p, r) in enumerate(rows):
        arr[k].offset_time = float(t)
        for i in range(3):
            arr[k].acc[i], arr[k].gyr[i], arr[k].vel[i], arr[k].pos[i] = float(a[i]), float(g[i]), float(v[i]), float(p[i])
        for i in range(9):
            arr[k].rot[i] = float(np.asarray(r).reshape(9)[i])
    return arr


def undistort(poses, x_end, pts_xyzt, first_point=True):
    """Per-point half of UndistortPcl: pts_xyzt = n x 4 float32 (x, y, z, time offset in ms).  first_point: reproduce the
    reference's repeated compensation of the earliest point (IMU_Processing.hpp:345), the default."""
    a = _c32(pts_xyzt).reshape(-1, 4)
    L = lib()
    L.orc_set_undistort_first.restype = None
    L.orc_set_undistort_first.argtypes = [C.c_int]
    L.orc_set_undistort_first(1 if first_point else 0)
    L.orc_undistort.restype = None
    L.orc_undistort.argtypes = [C.c_void_p, C.c_int, np.ctypeslib.ndpointer(np.float64), np.ctypeslib.ndpointer(np.float32),
                                C.c_size_t, C.c_size_t, C.c_size_t, np.ctypeslib.ndpointer(np.float32)]
    out = np.zeros((max(len(a), 1), 3), np.float32)
    L.orc_undistort(C.cast(poses, C.c_void_p), len(poses), _c64(x_end), a, 4, 3, len(a), out)
    L.orc_set_undistort_first(1)
    return out[: len(a)].copy()


class ImuState(C.Structure):
    """orc_imu_state: ImuProcess members that persist across scans (defaults of the reference's constructor, :115-128)."""
    _fields_ = [("mean_acc", C.c_double * 3), ("cov_acc", C.c_double * 3), ("cov_gyr", C.c_double * 3), ("cov_bias_gyr", C.c_double * 3),
                ("cov_bias_acc", C.c_double * 3), ("angvel_last", C.c_double * 3), ("acc_s_last", C.c_double * 3),
                ("last_imu", C.c_double * 7), ("last_lidar_end_time", C.c_double)]

    def __init__(self):
        super().__init__()
        self.mean_acc[:] = [0, 0, -1.0]
        self.cov_acc[:] = [0.1] * 3
        self.cov_gyr[:] = [0.1] * 3
        self.cov_bias_gyr[:] = [0.0001] * 3
        self.cov_bias_acc[:] = [0.0001] * 3


def imu_forward(st: ImuState, imu, pcl_beg_time, pcl_end_time, x, P):
    """Forward half of UndistortPcl.  imu: n x 7 (t, acc, gyr).  Returns (poses array, x_end, P_end)."""
    a = _c64(imu).reshape(-1, 7)
    L = lib()
    L.orc_imu_forward.restype = C.c_int
    L.orc_imu_forward.argtypes = [C.POINTER(ImuState), np.ctypeslib.ndpointer(np.float64), C.c_int, C.c_double, C.c_double,
                                  np.ctypeslib.ndpointer(np.float64), np.ctypeslib.ndpointer(np.float64), C.c_void_p]
    x = _c64(x).copy()
    P = _c64(P).reshape(NDOF, NDOF).copy()
    poses = (Pose6D * (len(a) + 1))()
    n = L.orc_imu_forward(C.byref(st), a, len(a), float(pcl_beg_time), float(pcl_end_time), x, P, C.cast(poses, C.c_void_p))
    out = (Pose6D * n)()
    C.memmove(out, poses, C.sizeof(Pose6D) * n)
    return out, x, P
