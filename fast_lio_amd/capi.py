"""ctypes binding of libfastlio_hip.so (include/fastlio_hip.h).

This is plumbing for tests and bench.py: every call goes straight through the C ABI a C++ node would
use.  There is no Python/CPU fallback -- if the library or a HIP device is missing, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _build

NSTATE = 26
NDOF = 23
K = 5

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


class FlhConfig(C.Structure):
    _fields_ = [
        ("device", C.c_int),
        ("cell_size", C.c_float),
        ("plane_threshold", C.c_float),
        ("max_sqdist", C.c_float),
        ("stream", C.c_void_p),
        ("lanes_per_query", C.c_int),
        ("sort_queries", C.c_int),
        ("pass_kernel", C.c_int),
        ("eigen_order", C.c_int),
        ("plane_fit_dtype", C.c_int),
        ("undistort_first_point", C.c_int),
        ("plane_cache", C.c_int),
        ("fused_small_changes", C.c_int),
        ("prelaunch", C.c_int),
        ("index_cache", C.c_int),
        ("stage_sort", C.c_int),
    ]


class FlhTiming(C.Structure):
    _fields_ = [("search_ms", C.c_float), ("fit_ms", C.c_float), ("total_ms", C.c_float), ("candidates", C.c_int64)]


class FlhMeas(C.Structure):
    _fields_ = [
        ("valid", C.c_int),
        ("n_eff", C.c_int64),
        ("has_normal_eq", C.c_int),
        ("HTH", C.c_double * 144),
        ("HTh", C.c_double * 12),
        ("h_x", C.POINTER(C.c_double)),
        ("h", C.POINTER(C.c_double)),
        ("total_residual", C.c_double),
    ]


class FlhUpdateStats(C.Structure):
    _fields_ = [
        ("passes", C.c_int),
        ("searches", C.c_int),
        ("returned_in_loop", C.c_int),
        ("n_eff", C.c_int * 8),
        ("pass_search", C.c_int * 8),
        ("pass_ms", C.c_double * 8),
        ("h_ms", C.c_double),
        ("solve_ms", C.c_double),
    ]


class FlhScanJob(C.Structure):
    _fields_ = [("pts", C.c_void_p), ("stride_bytes", C.c_size_t), ("N", C.c_size_t), ("x", C.c_void_p), ("P", C.c_void_p),
                ("slot", C.c_int)]


class FlhRunStats(C.Structure):
    _fields_ = [("scans", C.c_int64), ("passes", C.c_int64), ("searches", C.c_int64), ("n_search_passes", C.c_int64),
                ("n_nosearch_passes", C.c_int64), ("ms_search_passes", C.c_double), ("ms_nosearch_passes", C.c_double),
                ("ms_map_incremental", C.c_double)]


MEAS_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(FlhMeas))

# every symbol include/fastlio_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "flh_default_config", "flh_create", "flh_destroy", "flh_last_error", "flh_device_available", "flh_map_build",
    "flh_map_size", "flh_scan_upload", "flh_scan_size", "flh_scan_stage", "flh_scan_activate", "flh_get_counters", "flh_set_timing_stride",
    "flh_eval", "flh_eval_device", "flh_unpack_gram",
    "flh_fetch_selected", "flh_fetch_neighbors", "flh_fetch_world", "flh_fetch_normvec", "flh_fetch_rows",
    "flh_last_timing", "flh_enable_stats", "flh_time_kernel", "flh_esekf_create", "flh_esekf_destroy",
    "flh_esekf_set_meas_model", "flh_esekf_change_x", "flh_esekf_change_P", "flh_esekf_get_x", "flh_esekf_get_P",
    "flh_esekf_predict", "flh_esekf_update",
    "flh_map_add", "flh_map_delete_boxes", "flh_map_download", "flh_map_incremental", "flh_fetch_map_incremental",
    "flh_fov_segment", "flh_scan_stage_downsampled", "flh_fetch_scan",
    "flh_scan_stage_undistorted", "flh_esekf_update_scan", "flh_map_stats",
    "flh_scan_stage_async", "flh_scan_wait", "flh_host_alloc", "flh_host_free", "flh_frame_world", "flh_points_body_to_world",
    "flh_esekf_last_error", "flh_rccl_unique_id", "flh_rccl_init_rank", "flh_rccl_init_all", "flh_rccl_destroy", "flh_rccl_size",
    "flh_rccl_rank", "flh_eval_group", "flh_set_owned_interval", "flh_esekf_run_scans", "flh_get_search_counters", "flh_set_timing_sampling",
    "flh_map_sync", "flh_eval_begin", "flh_eval_end", "flh_debug_bounds", "flh_debug_pass_stamps", "flh_debug_scan_order", "flh_debug_search_redone", "flh_debug_stage_stats", "flh_peer_open", "flh_peer_init_all", "flh_peer_close", "flh_peer_size", "flh_peer_rank", "flh_get_pass_stats", "flh_map_change_stats",
    "flh_eval_expect_next", "flh_set_prelaunch", "flh_get_prelaunch_stats", "flh_map_storage_stats",
]

_lib = None


class FlhLocalMap(C.Structure):
    """LocalMap_Points + Localmap_Initialized (src/laserMapping.cpp:228-229)."""
    _fields_ = [("vertex_min", C.c_float * 3), ("vertex_max", C.c_float * 3), ("initialized", C.c_int)]


class FlhError(RuntimeError):
    pass


def rccl_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    _chk(lib().flh_rccl_unique_id(buf), "flh_rccl_unique_id")
    return buf.raw


def rccl_init_all(handles):
    """One process, one Handle per device: a common communicator (ncclCommInitAll); use eval_group afterwards."""
    arr = (C.c_void_p * len(handles))(*[h.ptr for h in handles])
    _chk(lib().flh_rccl_init_all(arr, len(handles)), "flh_rccl_init_all")


def peer_init_all(handles):
    """One process, several Handles: every handle's passes publish their group sums to every handle's granule buffer
    (flh_peer_init_all); use eval_group afterwards."""
    arr = (C.c_void_p * len(handles))(*[h._h for h in handles])
    _chk(lib().flh_peer_init_all(arr, len(handles)), "flh_peer_init_all")


def eval_group(handles, x, do_search: bool, ext: bool = False):
    arr = (C.c_void_p * len(handles))(*[h.ptr for h in handles])
    HTH = np.zeros(144)
    HTh = np.zeros(12)
    n = C.c_int64()
    tr = C.c_double()
    _chk(lib().flh_eval_group(arr, len(handles), np.ascontiguousarray(x, np.float64), int(do_search), int(ext), HTH, HTh,
                              C.byref(n), C.byref(tr)), "flh_eval_group")
    return HTH.reshape(12, 12), HTh, int(n.value), float(tr.value)


_pinned = []


def pinned_empty(shape, dtype=np.float32) -> np.ndarray:
    """A numpy array in page-locked host memory (flh_host_alloc): the DMA engine reads scans from it where they lie.
    The memory lives until the process ends (the arrays handed out here are few and long-lived)."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    p = lib().flh_host_alloc(max(n, 1))
    if not p:
        raise FlhError("flh_host_alloc: " + lib().flh_last_error().decode())
    buf = (C.c_char * max(n, 1)).from_address(p)
    _pinned.append(buf)
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)


def lib():
    """Load (building if needed) the shared library.  Raises if it cannot be built/loaded."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if _build.needs_build():
        path = _build.build()
    _lib = _declare(C.CDLL(path))
    return _lib


class using_library:
    """Context manager: every call of this module goes through another build of the same ABI while it is active -- the tests'
    way to run the filter of _build.build_reference_algebra() (whose flh_* calls resolve to the product library it links)."""

    def __init__(self, path):
        self._L = _declare(C.CDLL(path))

    def __enter__(self):
        global _lib
        lib()
        self._old, _lib = _lib, self._L
        return self._L

    def __exit__(self, *a):
        global _lib
        _lib = self._old


def _declare(L):
    """argument / result types of every entry point"""
    L.flh_last_error.restype = C.c_char_p
    L.flh_device_available.restype = C.c_int
    L.flh_default_config.argtypes = [C.POINTER(FlhConfig)]
    L.flh_create.argtypes = [C.POINTER(FlhConfig), C.POINTER(C.c_void_p)]
    L.flh_destroy.argtypes = [C.c_void_p]
    L.flh_map_build.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
    L.flh_map_size.restype = C.c_size_t
    L.flh_map_size.argtypes = [C.c_void_p]
    L.flh_scan_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
    L.flh_map_add.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_double]
    L.flh_map_delete_boxes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.flh_map_download.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.flh_map_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.flh_map_change_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.flh_map_incremental.argtypes = [C.c_void_p, _f64p, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_uint32)]
    L.flh_fetch_map_incremental.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.flh_fov_segment.argtypes = [C.c_void_p, C.POINTER(FlhLocalMap), _f64p, C.c_double, C.c_float, C.c_void_p,
                                  C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    L.flh_scan_size.restype = C.c_size_t
    L.flh_scan_size.argtypes = [C.c_void_p]
    L.flh_scan_stage.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t]
    L.flh_scan_activate.argtypes = [C.c_void_p, C.c_int]
    L.flh_scan_stage_async.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t]
    L.flh_scan_wait.argtypes = [C.c_void_p, C.c_int]
    L.flh_host_alloc.restype = C.c_void_p
    L.flh_host_alloc.argtypes = [C.c_size_t]
    L.flh_host_free.argtypes = [C.c_void_p]
    L.flh_frame_world.argtypes = [C.c_void_p, C.c_int, _f64p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.flh_points_body_to_world.argtypes = [C.c_void_p, _f64p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    L.flh_rccl_unique_id.argtypes = [C.c_char_p]
    L.flh_rccl_init_rank.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    L.flh_rccl_init_all.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.flh_rccl_destroy.argtypes = [C.c_void_p]
    L.flh_rccl_destroy.restype = None
    L.flh_map_sync.argtypes = [C.c_void_p]
    L.flh_debug_bounds.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.flh_debug_pass_stamps.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS"), C.c_size_t]
    L.flh_peer_open.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    L.flh_peer_init_all.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.flh_peer_close.argtypes = [C.c_void_p]
    L.flh_peer_close.restype = None
    L.flh_peer_size.argtypes = [C.c_void_p]
    L.flh_peer_rank.argtypes = [C.c_void_p]
    L.flh_map_storage_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.flh_eval_expect_next.argtypes = [C.c_void_p, C.c_int]
    L.flh_set_prelaunch.argtypes = [C.c_void_p, C.c_int]
    L.flh_get_prelaunch_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.flh_get_pass_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.flh_rccl_size.argtypes = [C.c_void_p]
    L.flh_rccl_rank.argtypes = [C.c_void_p]
    L.flh_eval_group.argtypes = [C.POINTER(C.c_void_p), C.c_int, _f64p, C.c_int, C.c_int, _f64p, _f64p, C.POINTER(C.c_int64),
                                 C.POINTER(C.c_double)]
    L.flh_set_owned_interval.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float]
    L.flh_esekf_run_scans.argtypes = [C.c_void_p, C.POINTER(FlhScanJob), C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_int,
                                      C.c_double, C.c_int, C.POINTER(FlhRunStats), C.c_void_p, C.c_void_p]
    L.flh_esekf_last_error.restype = C.c_char_p
    L.flh_esekf_last_error.argtypes = [C.c_void_p]
    L.flh_scan_stage_downsampled.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_float,
                                             C.POINTER(C.c_size_t)]
    L.flh_fetch_scan.argtypes = [C.c_void_p, C.c_void_p]
    L.flh_debug_scan_order.argtypes = [C.c_void_p, C.c_void_p]
    L.flh_debug_search_redone.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.flh_debug_stage_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    L.flh_scan_stage_undistorted.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                             C.c_int, _f64p, C.c_float, C.c_void_p, C.POINTER(C.c_size_t)]
    L.flh_get_counters.argtypes = [C.c_void_p, _f64p, C.c_int]
    L.flh_set_timing_stride.argtypes = [C.c_void_p, C.c_int]
    L.flh_get_search_counters.argtypes = [C.c_void_p, _f64p]
    L.flh_set_timing_sampling.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.flh_eval.argtypes = [C.c_void_p, _f64p, _f64p, _f64p, _f64p, C.c_int, C.c_int, _f64p, _f64p,
                           C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    L.flh_eval_device.argtypes = [C.c_void_p, _f64p, C.c_int, C.c_int, C.c_void_p]
    L.flh_unpack_gram.argtypes = [_f64p, _f64p, _f64p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    L.flh_fetch_selected.argtypes = [C.c_void_p, C.c_void_p]
    L.flh_fetch_neighbors.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.flh_fetch_world.argtypes = [C.c_void_p, C.c_void_p]
    L.flh_fetch_normvec.argtypes = [C.c_void_p, C.c_void_p]
    L.flh_fetch_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.flh_last_timing.argtypes = [C.c_void_p, C.POINTER(FlhTiming)]
    L.flh_enable_stats.argtypes = [C.c_void_p, C.c_int]
    L.flh_time_kernel.argtypes = [C.c_void_p, C.c_int, _f64p, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.flh_esekf_create.restype = C.c_void_p
    L.flh_esekf_create.argtypes = [C.c_void_p, C.c_int, _f64p, C.c_int]
    L.flh_esekf_destroy.argtypes = [C.c_void_p]
    L.flh_esekf_set_meas_model.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.flh_esekf_change_x.argtypes = [C.c_void_p, _f64p]
    L.flh_esekf_change_P.argtypes = [C.c_void_p, _f64p]
    L.flh_esekf_get_x.argtypes = [C.c_void_p, _f64p]
    L.flh_esekf_get_P.argtypes = [C.c_void_p, _f64p]
    L.flh_esekf_predict.argtypes = [C.c_void_p, C.c_double, _f64p, _f64p, _f64p]
    L.flh_esekf_update.argtypes = [C.c_void_p, C.c_double, C.POINTER(FlhUpdateStats)]
    L.flh_esekf_update_scan.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.POINTER(FlhUpdateStats)]
    _lib = L
    return L


def _chk(rc, what):
    if rc != 0:
        raise FlhError(f"{what}: {lib().flh_last_error().decode()}")


def device_available() -> bool:
    return bool(lib().flh_device_available())


class Handle:
    """flh_handle: the device-resident map + current scan."""

    def __init__(self, cell_size: float = 1.5, lanes_per_query: int = 4, device: int = -1, stream: int | None = None,
                 plane_threshold: float = 0.1, max_sqdist: float = 5.0, sort_queries: int = -1, pass_kernel: int = -1,
                 eigen_order: int = -1, plane_fit_dtype: int = 0, undistort_first_point: int = -1, plane_cache: int = -1,
                 fused_small_changes: int = -1, prelaunch: int = -1, index_cache: int = -1, stage_sort: int = -1):
        L = lib()
        cfg = FlhConfig()
        L.flh_default_config(C.byref(cfg))
        cfg.device = device
        cfg.cell_size = cell_size
        cfg.lanes_per_query = lanes_per_query
        cfg.plane_threshold = plane_threshold
        cfg.max_sqdist = max_sqdist
        cfg.stream = stream
        cfg.sort_queries = sort_queries
        cfg.pass_kernel = pass_kernel
        cfg.eigen_order = eigen_order
        cfg.plane_fit_dtype = plane_fit_dtype
        cfg.undistort_first_point = undistort_first_point
        cfg.plane_cache = plane_cache
        cfg.fused_small_changes = fused_small_changes
        cfg.prelaunch = prelaunch
        cfg.index_cache = index_cache
        cfg.stage_sort = stage_sort
        self._h = C.c_void_p()
        _chk(L.flh_create(C.byref(cfg), C.byref(self._h)), "flh_create")
        self._keep = []

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().flh_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ptr(self):
        return self._h

    def map_build(self, xyz: np.ndarray):
        a = np.ascontiguousarray(xyz, dtype=np.float32)
        assert a.ndim == 2 and a.shape[1] in (3, 4, 12)
        _chk(lib().flh_map_build(self._h, a.ctypes.data, a.shape[1] * 4, a.shape[0]), "flh_map_build")

    # ---- incremental map ----
    def map_add(self, xyz: np.ndarray, downsample: bool = True, downsample_size: float = 0.5):
        """ikdtree.Add_Points(points, downsample_on) -- src/laserMapping.cpp:470-471."""
        a = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        _chk(lib().flh_map_add(self._h, a.ctypes.data, 12, a.shape[0], int(downsample), float(downsample_size)),
             "flh_map_add")

    def map_delete_boxes(self, boxes: np.ndarray):
        """ikdtree.Delete_Point_Boxes -- src/laserMapping.cpp:275.  boxes: nb x (min xyz, max xyz)."""
        b = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 6)
        _chk(lib().flh_map_delete_boxes(self._h, b.ctypes.data, b.shape[0]), "flh_map_delete_boxes")

    def map_stats(self) -> dict:
        out = (C.c_uint64 * 6)()
        _chk(lib().flh_map_stats(self._h, out), "flh_map_stats")
        return {"reindex": int(out[0]), "brickwise": int(out[1]), "slots_used": int(out[2]), "slots": int(out[3]),
                "ids": int(out[4]), "bricks": int(out[5])}

    def map_storage_stats(self) -> dict:
        out = (C.c_uint64 * 4)()
        _chk(lib().flh_map_storage_stats(self._h, out), "flh_map_storage_stats")
        return {"points": int(out[0]), "slots_in_brick_ranges": int(out[1]), "bricks_purged": int(out[2]), "bricks": int(out[3])}

    def map_change_stats(self) -> dict:
        out = (C.c_uint64 * 2)()
        _chk(lib().flh_map_change_stats(self._h, out), "flh_map_change_stats")
        return {"enqueued_without_wait": int(out[0]), "replayed": int(out[1])}

    def search_redone(self) -> int:
        """Searching passes repeated because the map change they were enqueued behind asked for a re-index / a replay."""
        out = C.c_uint64(0)
        _chk(lib().flh_debug_search_redone(self._h, C.byref(out)), "flh_debug_search_redone")
        return int(out.value)

    def stage_stats(self, reset=False) -> dict:
        """Developer counters of the staging / activation hand-over -- flh_debug_stage_stats."""
        out = (C.c_double * 10)()
        _chk(lib().flh_debug_stage_stats(self._h, out, 1 if reset else 0), "flh_debug_stage_stats")
        k = ("jobs", "enq_us", "enq_max_us", "h2d_wait_us", "h2d_wait_max_us", "activations", "act_wait_us", "act_wait_max_us",
             "act_event_not_ready", "act_slot_pending")
        return {n: float(out[i]) for i, n in enumerate(k)}

    def map_download(self) -> np.ndarray:
        out = np.zeros((self.M, 3), np.float32)
        _chk(lib().flh_map_download(self._h, out.ctypes.data, out.shape[0]), "flh_map_download")
        return out

    def map_incremental(self, x, filter_size_map: float = 0.5, flg_EKF_inited: bool = True, apply: bool = True, counts: bool = True):
        """map_incremental() -- src/laserMapping.cpp:427-474.  Returns (n_add, n_no_downsample); counts=False: nobody asks for the
        two list lengths (as the node's loop does not), which lets the library enqueue Add_Points without waiting for them -- None."""
        if not counts:
            _chk(lib().flh_map_incremental(self._h, np.ascontiguousarray(x, dtype=np.float64), float(filter_size_map),
                                           int(flg_EKF_inited), int(apply), None, None), "flh_map_incremental")
            return None
        n1, n2 = C.c_uint32(0), C.c_uint32(0)
        _chk(lib().flh_map_incremental(self._h, np.ascontiguousarray(x, dtype=np.float64), float(filter_size_map),
                                       int(flg_EKF_inited), int(apply), C.byref(n1), C.byref(n2)), "flh_map_incremental")
        return int(n1.value), int(n2.value)

    def fetch_map_incremental(self):
        cls = np.zeros(self.N, np.uint8)
        world = np.zeros((self.N, 3), np.float32)
        _chk(lib().flh_fetch_map_incremental(self._h, cls.ctypes.data, world.ctypes.data), "flh_fetch_map_incremental")
        return world, cls

    def fov_segment(self, lm: "FlhLocalMap", pos_lid, cube_len: float = 200.0, det_range: float = 300.0):
        """lasermap_fov_segment() -- src/laserMapping.cpp:230-280.  Returns (boxes nb x 6, points deleted)."""
        boxes = np.zeros((3, 6), np.float32)
        nb, ndel = C.c_int(0), C.c_int64(0)
        _chk(lib().flh_fov_segment(self._h, C.byref(lm), np.ascontiguousarray(pos_lid, dtype=np.float64), float(cube_len),
                                   float(det_range), boxes.ctypes.data, C.byref(nb), C.byref(ndel)), "flh_fov_segment")
        return boxes[: nb.value].copy(), int(ndel.value)

    def scan_upload(self, body: np.ndarray):
        a = np.ascontiguousarray(body, dtype=np.float32)
        assert a.ndim == 2 and a.shape[1] in (3, 4, 12)
        _chk(lib().flh_scan_upload(self._h, a.ctypes.data, a.shape[1] * 4, a.shape[0]), "flh_scan_upload")

    def scan_stage(self, slot: int, body: np.ndarray):
        a = np.ascontiguousarray(body, dtype=np.float32)
        _chk(lib().flh_scan_stage(self._h, slot, a.ctypes.data, a.shape[1] * 4, a.shape[0]), "flh_scan_stage")

    # ---- multi-GPU ----
    def rccl_init_rank(self, nranks: int, unique_id: bytes, rank: int):
        """Join the RCCL communicator (one process per GPU); afterwards eval / Esekf.update all-reduce every pass."""
        assert len(unique_id) == 128
        _chk(lib().flh_rccl_init_rank(self._h, nranks, unique_id, rank), "flh_rccl_init_rank")

    def peer_open(self, shm_name: str, nranks: int, rank: int):
        """One process per GPU: attach to the ranks' shared granule segment (flh_peer_open); rank 0 creates it."""
        _chk(lib().flh_peer_open(self._h, shm_name.encode(), nranks, rank), "flh_peer_open")

    def peer_close(self):
        lib().flh_peer_close(self._h)

    def peer_size(self) -> int:
        return int(lib().flh_peer_size(self._h))

    def peer_rank(self) -> int:
        return int(lib().flh_peer_rank(self._h))

    def debug_bounds(self):
        """(instrumented?, [20 words]) -- see flh_debug_bounds."""
        out = (C.c_uint64 * 20)()
        rc = lib().flh_debug_bounds(self._h, out)
        if rc < 0:
            raise FlhError(lib().flh_last_error().decode())
        return bool(rc), [int(v) for v in out]

    def debug_pass_stamps(self, n_waves: int):
        """(instrumented?, n_waves x 12 array: 8 stamps in 100 MHz ticks, HW_ID, XCC_ID, longest candidate list, open queries)
        -- see flh_debug_pass_stamps (tools/pass_stamps.py)."""
        out = np.zeros((n_waves, 12), np.uint64)
        rc = lib().flh_debug_pass_stamps(self._h, out, out.size)
        if rc < 0:
            raise FlhError(lib().flh_last_error().decode())
        return bool(rc), out

    def pass_stats(self) -> dict:
        out = (C.c_uint64 * 4)()
        _chk(lib().flh_get_pass_stats(self._h, out), "flh_get_pass_stats")
        return {"search_passes": int(out[0]), "one_launch_passes": int(out[1]), "second_stage_queries": int(out[2]),
                "nosearch_passes": int(out[3])}

    def expect_next(self, kind: int) -> None:
        """flh_eval_expect_next: 0 unknown, 1 a no-search evaluation probably follows the next one, 2 nothing follows."""
        _chk(lib().flh_eval_expect_next(self._h, kind), "flh_eval_expect_next")

    def set_prelaunch(self, on: bool) -> None:
        _chk(lib().flh_set_prelaunch(self._h, 1 if on else 0), "flh_set_prelaunch")

    def prelaunch_stats(self) -> dict:
        out = (C.c_uint64 * 4)()
        _chk(lib().flh_get_prelaunch_stats(self._h, out), "flh_get_prelaunch_stats")
        return {"armed": int(out[0]), "go": int(out[1]), "abort": int(out[2]), "gone": int(out[3])}

    def rccl_size(self) -> int:
        return int(lib().flh_rccl_size(self._h))

    def set_owned_interval(self, axis: int, lo: float = 0.0, hi: float = 0.0):
        """Map partitioned over ranks: search only the queries whose world coordinate `axis` lies in [lo, hi); axis < 0 = all."""
        _chk(lib().flh_set_owned_interval(self._h, axis, float(lo), float(hi)), "flh_set_owned_interval")

    def scan_stage_async(self, slot: int, body: np.ndarray):
        """Staging by the handle's staging thread; `body` (float32, C-contiguous, N x 3/4/12) is kept alive here until
        scan_wait / scan_activate / update_scan of that slot."""
        a = body
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"] and a.ndim == 2
        self._pending = getattr(self, "_pending", {})
        self._pending[slot] = a
        _chk(lib().flh_scan_stage_async(self._h, slot, a.ctypes.data, a.shape[1] * 4, a.shape[0]), "flh_scan_stage_async")

    def scan_wait(self, slot: int):
        _chk(lib().flh_scan_wait(self._h, slot), "flh_scan_wait")
        getattr(self, "_pending", {}).pop(slot, None)

    def frame_world(self, x, slot: int = -1, dense: bool = True) -> np.ndarray:
        """publish_frame_world's cloud (src/laserMapping.cpp:478-530): feats_undistort (dense) or feats_down_body of
        `slot` (-1 = the active scan) carried to the world frame at state x."""
        n = C.c_size_t(0)
        xx = np.ascontiguousarray(x, np.float64)
        _chk(lib().flh_frame_world(self._h, slot, xx, int(dense), None, 0, C.byref(n)), "flh_frame_world")
        out = np.empty((n.value, 3), np.float32)
        if n.value:
            _chk(lib().flh_frame_world(self._h, slot, xx, int(dense), out.ctypes.data, n.value, C.byref(n)), "flh_frame_world")
        return out

    def points_body_to_world(self, x, pts: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(pts, dtype=np.float32)
        out = np.empty((a.shape[0], 3), np.float32)
        _chk(lib().flh_points_body_to_world(self._h, np.ascontiguousarray(x, np.float64), a.ctypes.data, a.shape[1] * 4,
                                            a.shape[0], out.ctypes.data), "flh_points_body_to_world")
        return out

    def scan_stage_downsampled(self, slot: int, raw: np.ndarray, leaf_size: float = 0.5) -> int:
        """pcl::VoxelGrid of the raw scan on the device + staging (src/laserMapping.cpp:904-905).  Returns feats_down_size."""
        a = np.ascontiguousarray(raw, dtype=np.float32)
        n_out = C.c_size_t(0)
        _chk(lib().flh_scan_stage_downsampled(self._h, slot, a.ctypes.data, a.shape[1] * 4 if a.ndim == 2 else 12,
                                              a.shape[0], float(leaf_size), C.byref(n_out)), "flh_scan_stage_downsampled")
        return int(n_out.value)

    def scan_stage_undistorted(self, slot: int, pts_xyzt: np.ndarray, poses, x_end, leaf_size: float = 0.5,
                               want_undistorted: bool = True):
        """UndistortPcl's per-point half + VoxelGrid + staging.  pts_xyzt: n x 4 float32 (x, y, z, time offset in ms);
        poses: a ctypes array of 22-double Pose6D records.  Returns (feats_down_size, feats_undistort n x 3 or None --
        the cloud stays on the device either way, see frame_world)."""
        a = pts_xyzt if (pts_xyzt.dtype == np.float32 and pts_xyzt.flags["C_CONTIGUOUS"]) else np.ascontiguousarray(pts_xyzt, dtype=np.float32)
        a = a.reshape(-1, 4)
        und = np.zeros((max(len(a), 1), 3), np.float32) if want_undistorted else None
        n_out = C.c_size_t(0)
        _chk(lib().flh_scan_stage_undistorted(self._h, slot, a.ctypes.data, 16, 12, a.shape[0], C.cast(poses, C.c_void_p),
                                              len(poses), np.ascontiguousarray(x_end, dtype=np.float64), float(leaf_size),
                                              und.ctypes.data if want_undistorted else None, C.byref(n_out)),
             "flh_scan_stage_undistorted")
        return int(n_out.value), (und[: len(a)].copy() if want_undistorted else None)

    def scan_order(self) -> np.ndarray:
        """order[i] = original index of the scan point at internal (Morton) position i -- flh_debug_scan_order."""
        out = np.zeros(self.N, np.uint32)
        _chk(lib().flh_debug_scan_order(self._h, out.ctypes.data), "flh_debug_scan_order")
        return out

    def fetch_scan(self) -> np.ndarray:
        out = np.zeros((self.N, 3), np.float32)
        _chk(lib().flh_fetch_scan(self._h, out.ctypes.data), "flh_fetch_scan")
        return out

    def scan_activate(self, slot: int):
        _chk(lib().flh_scan_activate(self._h, slot), "flh_scan_activate")

    def counters(self, reset: bool = False):
        out = np.zeros(6)
        _chk(lib().flh_get_counters(self._h, out, int(reset)), "flh_get_counters")
        return {"search_ms": out[0], "n_search": int(out[1]), "fit_ms": out[2], "n_fit": int(out[3]),
                "eval_ms": out[4], "n_eval": int(out[5])}

    def search_counters(self):
        """Search-kernel time split by kind: a scan's first search / its later (cache-bounded) searches."""
        out = np.zeros(4)
        _chk(lib().flh_get_search_counters(self._h, out), "flh_get_search_counters")
        return {"first_ms": out[0], "n_first": int(out[1]), "later_ms": out[2], "n_later": int(out[3])}

    @property
    def N(self):
        return int(lib().flh_scan_size(self._h))

    @property
    def M(self):
        return int(lib().flh_map_size(self._h))

    def eval(self, x: np.ndarray, do_search: bool, extrinsic_est_en: bool = False):
        x = np.ascontiguousarray(x, np.float64)
        HTH = np.zeros(144)
        HTh = np.zeros(12)
        n = C.c_int64()
        tr = C.c_double()
        _chk(lib().flh_eval(self._h, np.ascontiguousarray(x[3:7]), np.ascontiguousarray(x[0:3]),
                            np.ascontiguousarray(x[7:11]), np.ascontiguousarray(x[11:14]), int(do_search),
                            int(extrinsic_est_en), HTH, HTh, C.byref(n), C.byref(tr)), "flh_eval")
        return HTH.reshape(12, 12), HTh, int(n.value), float(tr.value)

    def eval_device(self, x: np.ndarray, do_search: bool, extrinsic_est_en: bool, d_gram_ptr: int):
        _chk(lib().flh_eval_device(self._h, np.ascontiguousarray(x, np.float64), int(do_search), int(extrinsic_est_en),
                                   C.c_void_p(d_gram_ptr)), "flh_eval_device")

    def fetch_selected(self):
        out = np.zeros(self.N, np.uint8)
        _chk(lib().flh_fetch_selected(self._h, out.ctypes.data), "flh_fetch_selected")
        return out

    def fetch_neighbors(self):
        n = self.N
        idx = np.zeros((n, K), np.int32)
        d2 = np.zeros((n, K), np.float32)
        cnt = np.zeros(n, np.uint8)
        _chk(lib().flh_fetch_neighbors(self._h, idx.ctypes.data, d2.ctypes.data, cnt.ctypes.data), "flh_fetch_neighbors")
        return idx, d2, cnt

    def fetch_world(self):
        out = np.zeros((self.N, 3), np.float32)
        _chk(lib().flh_fetch_world(self._h, out.ctypes.data), "flh_fetch_world")
        return out

    def fetch_normvec(self):
        out = np.zeros((self.N, 4), np.float32)
        _chk(lib().flh_fetch_normvec(self._h, out.ctypes.data), "flh_fetch_normvec")
        return out

    def fetch_rows(self):
        n = C.c_int64()
        _chk(lib().flh_fetch_rows(self._h, None, None, 0, C.byref(n)), "flh_fetch_rows")
        rows = int(n.value)
        hx = np.zeros((12, max(rows, 1)))
        hv = np.zeros(max(rows, 1))
        _chk(lib().flh_fetch_rows(self._h, hx.ctypes.data, hv.ctypes.data, rows, C.byref(n)), "flh_fetch_rows")
        return np.ascontiguousarray(hx[:, :rows].T), hv[:rows]

    def timing(self):
        t = FlhTiming()
        _chk(lib().flh_last_timing(self._h, C.byref(t)), "flh_last_timing")
        return {"search_ms": t.search_ms, "fit_ms": t.fit_ms, "total_ms": t.total_ms, "candidates": int(t.candidates)}

    def set_timing_stride(self, every_n: int):
        _chk(lib().flh_set_timing_stride(self._h, int(every_n)), "flh_set_timing_stride")

    def set_timing_sampling(self, every_n: int, search_only: bool):
        _chk(lib().flh_set_timing_sampling(self._h, int(every_n), int(search_only)), "flh_set_timing_sampling")

    def enable_stats(self, on=True):
        _chk(lib().flh_enable_stats(self._h, int(on)), "flh_enable_stats")

    def time_kernel(self, which: int, x: np.ndarray, extrinsic_est_en: bool = False, iters: int = 20) -> float:
        ms = C.c_float()
        _chk(lib().flh_time_kernel(self._h, which, np.ascontiguousarray(x, np.float64), int(extrinsic_est_en), iters,
                                   C.byref(ms)), "flh_time_kernel")
        return float(ms.value)


class Esekf:
    """flh_esekf: the host IEKF (C++ mirror of esekfom::esekf<state_ikfom,12,input_ikfom>)."""

    def __init__(self, handle: Handle | None, max_iter: int = 3, limit=None, extrinsic_est_en: bool = False):
        L = lib()
        lim = np.ascontiguousarray(np.full(NDOF, 0.001) if limit is None else limit, np.float64)
        self._handle = handle
        self._e = C.c_void_p(L.flh_esekf_create(handle.ptr if handle is not None else None, max_iter, lim,
                                                int(extrinsic_est_en)))
        self._cb = None

    def close(self):
        if getattr(self, "_e", None) and self._e.value:
            lib().flh_esekf_destroy(self._e)
            self._e = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_meas_model(self, fn):
        """fn(x: np.ndarray[26], converge: bool) -> dict(valid, n_eff, HTH(12x12)|None, HTh|None, h_x(n x12)|None, h|None)."""
        if fn is None:
            self._cb = None
            lib().flh_esekf_set_meas_model(self._e, None, None)
            return
        keep = {}

        def _tramp(ctx, xptr, converge, out):
            x = np.ctypeslib.as_array(xptr, shape=(NSTATE,)).copy()
            r = fn(x, bool(converge))
            o = out.contents
            o.valid = int(bool(r.get("valid", True)))
            o.n_eff = int(r.get("n_eff", 0))
            o.total_residual = float(r.get("total_residual", 0.0))
            if r.get("HTH") is not None:
                o.has_normal_eq = 1
                C.memmove(o.HTH, np.ascontiguousarray(r["HTH"], np.float64).ctypes.data, 144 * 8)
                C.memmove(o.HTh, np.ascontiguousarray(r["HTh"], np.float64).ctypes.data, 12 * 8)
            else:
                o.has_normal_eq = 0
            if r.get("h_x") is not None and o.n_eff > 0:
                keep["hx"] = np.ascontiguousarray(np.asarray(r["h_x"], np.float64).T)  # -> column-major n x 12
                keep["h"] = np.ascontiguousarray(r["h"], np.float64)
                o.h_x = keep["hx"].ctypes.data_as(C.POINTER(C.c_double))
                o.h = keep["h"].ctypes.data_as(C.POINTER(C.c_double))
            else:
                o.h_x = None
                o.h = None

        self._cb = MEAS_FN(_tramp)
        lib().flh_esekf_set_meas_model(self._e, C.cast(self._cb, C.c_void_p), None)

    def change_x(self, x):
        lib().flh_esekf_change_x(self._e, np.ascontiguousarray(x, np.float64))

    def change_P(self, P):
        lib().flh_esekf_change_P(self._e, np.ascontiguousarray(P, np.float64).reshape(-1))

    def get_x(self):
        x = np.zeros(NSTATE)
        lib().flh_esekf_get_x(self._e, x)
        return x

    def get_P(self):
        P = np.zeros(NDOF * NDOF)
        lib().flh_esekf_get_P(self._e, P)
        return P.reshape(NDOF, NDOF)

    def predict(self, dt, Q, acc, gyro):
        lib().flh_esekf_predict(self._e, float(dt), np.ascontiguousarray(Q, np.float64).reshape(-1),
                                np.ascontiguousarray(acc, np.float64), np.ascontiguousarray(gyro, np.float64))

    def update_scan(self, slot: int, x: np.ndarray, P: np.ndarray, R: float = 0.001):
        """One loop body of the node: activate the staged scan, set the propagated (x, P), update.  x and P must be
        C-contiguous float64 arrays that stay alive during the call."""
        st = FlhUpdateStats()
        rc = lib().flh_esekf_update_scan(self._e, int(slot), x.ctypes.data, P.ctypes.data, float(R), C.byref(st))
        if rc != 0:
            raise FlhError("flh_esekf_update_scan failed: " + lib().flh_esekf_last_error(self._e).decode())
        return st

    @staticmethod
    def make_jobs(bodies, priors, slots=None):
        """ctypes array of flh_scan_job for run_scans: bodies = list of float32 C-contiguous N x 3/4 arrays (kept alive by the
        caller), priors = list of (x, P) float64 C-contiguous arrays, slots = None (stage from the buffers) or a list of slots."""
        arr = (FlhScanJob * len(bodies))()
        for i, (b, (x, P)) in enumerate(zip(bodies, priors)):
            assert b.dtype == np.float32 and b.flags["C_CONTIGUOUS"] and x.dtype == np.float64 and P.dtype == np.float64
            arr[i].pts = b.ctypes.data
            arr[i].stride_bytes = b.shape[1] * 4
            arr[i].N = b.shape[0]
            arr[i].x = x.ctypes.data
            arr[i].P = P.ctypes.data
            arr[i].slot = -1 if slots is None else int(slots[i])
        return arr

    def run_scans(self, jobs, first: int, count: int, ring: int = 4, R: float = 0.001, map_incremental: bool = False,
                  filter_size_map: float = 0.5, first_staged: bool = False, stage_next: bool = False) -> FlhRunStats:
        """The node's main loop over `count` scans, natively (flh_esekf_run_scans): no Python between the scans.
        stage_next / first_staged chain two calls into one continuous stream (the scan after this call's last is staged while
        the last one updates)."""
        rs = FlhRunStats()
        rc = lib().flh_esekf_run_scans(self._e, jobs, len(jobs), int(first), int(count), int(ring), float(R), int(map_incremental),
                                       float(filter_size_map), (1 if first_staged else 0) | (2 if stage_next else 0),
                                       C.byref(rs), None, None)
        if rc != 0:
            raise FlhError("flh_esekf_run_scans failed: " + lib().flh_esekf_last_error(self._e).decode())
        return rs

    def update(self, R: float = 0.001):
        st = FlhUpdateStats()
        rc = lib().flh_esekf_update(self._e, float(R), C.byref(st))
        if rc != 0:
            raise FlhError("flh_esekf_update failed: " + lib().flh_esekf_last_error(self._e).decode())
        return st


def predict_fn(x, P, dt, Q, acc, gyro):
    """esekf::predict through the product's host library (for synth.propagate_prior_cov)."""
    kf = Esekf(None)
    kf.change_x(x)
    kf.change_P(P)
    kf.predict(dt, Q, acc, gyro)
    out = kf.get_x(), kf.get_P()
    kf.close()
    return out
