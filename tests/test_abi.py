"""The C-ABI library loads on a CPU-only box, exports every symbol include/fastlio_hip.h declares, and
refuses to compute without a GPU (no silent CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from fast_lio_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    # the drop-in boundary + the developer header (instrumentation readers, stubs in the product build)
    hdr = open(os.path.join(ROOT, "include", "fastlio_hip.h")).read() + open(os.path.join(ROOT, "include", "fastlio_hip_dev.h")).read()
    declared = sorted(set(re.findall(r"\b(flh_[a-z_A-Z0-9]+)\s*\(", hdr)))
    declared = [d for d in declared if d not in ("flh_meas_fn",)]
    assert set(declared) == set(capi.EXPORTS), set(declared) ^ set(capi.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_unpack_gram_layout():
    G = np.arange(256, dtype=np.float64)
    HTH = np.zeros(144)
    HTh = np.zeros(12)
    n = C.c_int64()
    tr = C.c_double()
    capi.lib().flh_unpack_gram(G, HTH, HTh, C.byref(n), C.byref(tr))
    Gm = G.reshape(16, 16)
    np.testing.assert_array_equal(HTH.reshape(12, 12), Gm[:12, :12])
    np.testing.assert_array_equal(HTh, Gm[:12, 12])
    assert n.value == int(Gm[13, 13]) and tr.value == Gm[14, 13]


def test_no_cpu_fallback_without_gpu():
    if capi.device_available():
        pytest.skip("a GPU is visible; the loud-failure path is for CPU-only boxes")
    with pytest.raises(capi.FlhError, match="no HIP device"):
        capi.Handle()


def test_config_mirror_matches_the_c_struct():
    """capi.FlhConfig must lay out exactly as `flh_config` in include/fastlio_hip.h: same fields in the same order (parsed from the
    header), and flh_default_config's values land in the fields they belong to (a shifted field would show up here, on a
    CPU-only box, instead of as a wrong kernel choice on the GPU)."""
    hdr = open(os.path.join(ROOT, "include", "fastlio_hip.h")).read()
    body = hdr[hdr.index("typedef struct flh_config {"):hdr.index("} flh_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(?:int|float|void\s*\*)\s*([a-z_0-9]+)\s*;", body)
    assert fields == [f for f, _ in capi.FlhConfig._fields_], (fields, [f for f, _ in capi.FlhConfig._fields_])
    cfg = capi.FlhConfig()
    C.memset(C.byref(cfg), 0x5A, C.sizeof(cfg))
    capi.lib().flh_default_config(C.byref(cfg))
    assert cfg.device == -1 and cfg.lanes_per_query == 4 and cfg.plane_fit_dtype == 0
    assert cfg.cell_size == 1.5 and cfg.max_sqdist == 5.0 and abs(cfg.plane_threshold - 0.1) < 1e-7
    assert not cfg.stream
    # "default" markers the library resolves in flh_create
    assert (cfg.sort_queries, cfg.pass_kernel, cfg.eigen_order, cfg.undistort_first_point, cfg.plane_cache, cfg.fused_small_changes,
            cfg.prelaunch, cfg.index_cache, cfg.stage_sort) == (-1,) * 9
