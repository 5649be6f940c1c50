"""The scan's staging by the library's own two kernels (fast_lio_amd/csrc/flh_stage.hip: tile sort in LDS + merge by rank) against
(a) numpy's stable argsort of the very key the kernels form and (b) the vendor library's radix sort the staging used to call
(flh_config.stage_sort = 0): the same order element for element -- hence the same 64-point units, the same summation tree, the
same bits in everything downstream.  The hand-over the staging serves: src/laserMapping.cpp:904-905, 935-951."""
import numpy as np
import pytest

from fast_lio_amd import capi, synth

pytestmark = pytest.mark.gpu

TILE = 4096  # flh_stage.hip: records per tile up to 131 072 points (8 192 above)


def morton_key(xyz: np.ndarray) -> np.ndarray:
    """flh_device.hpp scan_morton at a 0.5 m quantum, in numpy (fp32 arithmetic, the same clamps)."""
    def spread(v):
        x = v & np.uint32(0x3FF)
        x = (x | (x << np.uint32(16))) & np.uint32(0x030000FF)
        x = (x | (x << np.uint32(8))) & np.uint32(0x0300F00F)
        x = (x | (x << np.uint32(4))) & np.uint32(0x030C30C3)
        x = (x | (x << np.uint32(2))) & np.uint32(0x09249249)
        return x
    p = np.asarray(xyz, np.float32)
    two = np.float32(2.0)
    ix = np.minimum(np.maximum(p[:, 0] * two + np.float32(1024.0), np.float32(0.0)), np.float32(2047.0)).astype(np.uint32)
    iy = np.minimum(np.maximum(p[:, 1] * two + np.float32(1024.0), np.float32(0.0)), np.float32(2047.0)).astype(np.uint32)
    iz = np.minimum(np.maximum(p[:, 2] * two + np.float32(512.0), np.float32(0.0)), np.float32(1023.0)).astype(np.uint32)
    return (spread(ix) | (spread(iy) << np.uint32(1)) | (spread(iz) << np.uint32(2)) | ((ix >> np.uint32(10)) << np.uint32(30))
            | ((iy >> np.uint32(10)) << np.uint32(31)))


@pytest.fixture(scope="module")
def handles():
    pr = synth.make_problem(200000, 20000, "avia", cfg=1)
    own = capi.Handle(stage_sort=2)   # (2: the library's own kernels wherever they can run -- 262 144 points; 1, the default: 114 688)
    lib = capi.Handle(stage_sort=0)
    for h in (own, lib):
        h.map_build(pr.map_xyz)
    yield pr, own, lib
    own.close()
    lib.close()


def clouds(pr):
    rng = np.random.default_rng(11)
    big = np.ascontiguousarray(np.tile(pr.body, (14, 1)) + rng.normal(0, 0.3, (14 * len(pr.body), 3)).astype(np.float32))
    lattice = np.ascontiguousarray((rng.integers(-6, 6, (30000, 3)) * 0.5 + 0.1).astype(np.float32))   # few distinct keys: ties
    far = (rng.uniform(-600, 600, (20000, 3))).astype(np.float32)                                        # every clamp of the key
    same = np.ascontiguousarray(np.tile(np.float32([[1.0, 2.0, 3.0]]), (3 * TILE + 5, 1)))                # ONE key: pure stability
    out = {"tiny": pr.body[:7], "two": pr.body[:2], "one_tile_ragged": pr.body[:3003], "tile": big[:TILE], "tile+1": big[:TILE + 1],
           "two_tiles-1": big[:2 * TILE - 1], "20k": pr.body, "100k": big[:100000], "130k": big[:130000], "small_max": big[:131072],
           "large_min": big[:131073], "200k": big[:200000], "max": big[:262144], "lattice": lattice, "far": far, "same": same,
           "skew": np.concatenate([big[:9000][np.argsort(morton_key(big[:9000]), kind="stable")], big[9000:30000]])}
    return {k: np.ascontiguousarray(v) for k, v in out.items()}, big


def test_own_staging_is_the_stable_order_of_the_key(handles):
    pr, own, _ = handles
    cl, _ = clouds(pr)
    for name, body in cl.items():
        own.scan_upload(body)
        got = own.scan_order()
        want = np.argsort(morton_key(body), kind="stable").astype(np.uint32)
        np.testing.assert_array_equal(got, want, err_msg=name)
        np.testing.assert_array_equal(own.fetch_scan().view(np.uint32), body.view(np.uint32), err_msg=name)


def test_own_staging_equals_library_sort(handles):
    """Order, and through it the normal equations of a searching and a no-search pass, bit for bit; every record layout the
    staging accepts (12-, 16- and 48-byte strides); the staging thread; a scan larger than the own path takes (falls back)."""
    pr, own, lib = handles
    cl, big = clouds(pr)
    xp, _ = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
    for name in ("tiny", "one_tile_ragged", "tile+1", "20k", "100k", "lattice"):
        body = cl[name]
        res = []
        for h in (own, lib):
            h.scan_upload(body)
            a = h.eval(xp, True, False)
            b = h.eval(pr.x_true, False, False)
            res.append((h.scan_order(), a, b, h.fetch_selected()))
        np.testing.assert_array_equal(res[0][0], res[1][0], err_msg=name)
        for k in (1, 2):
            np.testing.assert_array_equal(res[0][k][0], res[1][k][0], err_msg=name)
            np.testing.assert_array_equal(res[0][k][1], res[1][k][1], err_msg=name)
            assert res[0][k][2:] == res[1][k][2:], name
        np.testing.assert_array_equal(res[0][3], res[1][3], err_msg=name)
    base = cl["100k"]
    for width in (3, 4, 12):  # xyz first, the rest of the record is whatever the caller keeps there
        rec = np.full((len(base), width), 7.25, np.float32)
        rec[:, :3] = base
        orders = []
        for h in (own, lib):
            h.scan_stage_async(2, rec)
            h.scan_activate(2)
            orders.append(h.scan_order())
            np.testing.assert_array_equal(h.fetch_scan().view(np.uint32), base.view(np.uint32))
        np.testing.assert_array_equal(orders[0], orders[1], err_msg=f"stride {4 * width}")
    over = np.ascontiguousarray(big[:262144 + 9])
    own.scan_upload(over)
    lib.scan_upload(over)
    np.testing.assert_array_equal(own.scan_order(), lib.scan_order())


def test_own_staging_behind_the_scan_front_end(handles):
    """flh_scan_stage_downsampled hands float4 points to the same two kernels (stride 16)."""
    pr, own, lib = handles
    rng = np.random.default_rng(5)
    raw = np.ascontiguousarray(np.tile(pr.body, (6, 1)) + rng.normal(0, 0.2, (6 * len(pr.body), 3)).astype(np.float32))
    got = []
    for h in (own, lib):
        n = h.scan_stage_downsampled(1, raw, 0.5)
        h.scan_activate(1)
        got.append((n, h.scan_order(), h.fetch_scan()))
    assert got[0][0] == got[1][0] and got[0][0] > 1000
    np.testing.assert_array_equal(got[0][1], got[1][1])
    np.testing.assert_array_equal(got[0][2].view(np.uint32), got[1][2].view(np.uint32))
