"""Known-answer tests for the oracle's pcl::VoxelGrid restatement (oracle_path.c: orc_voxel_grid), the stand-in for
downSizeFilterSurf.filter (src/laserMapping.cpp:904-905).  PCL is an external dependency of the reference; these pin the
published algorithm: voxel coordinates relative to floor(min * inv_leaf), linear index, centroids in index order."""
import numpy as np

from oracle import pyoracle as po


def ref_numpy(a, leaf):
    a = np.asarray(a, np.float32)
    inv = np.float32(1.0) / np.float32(leaf)
    mn_b = np.floor(a.min(0) * inv).astype(np.int64)
    div = np.floor(a.max(0) * inv).astype(np.int64) - mn_b + 1
    ijk = (np.floor(a * inv) - mn_b.astype(np.float32)).astype(np.int64)
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    out = []
    for k in np.unique(idx):
        s = np.zeros(3, np.float32)
        pts = a[idx == k]              # ascending input index
        for p in pts:
            s = s + p
        out.append(s / np.float32(len(pts)))
    return np.array(out, np.float32).reshape(-1, 3)


def test_single_voxel_is_the_float_mean():
    a = np.array([[0.1, 0.1, 0.1], [0.2, 0.3, 0.4], [0.45, 0.05, 0.25]], np.float32)
    out = po.voxel_grid(a, 0.5)
    s = (a[0] + a[1]) + a[2]
    np.testing.assert_array_equal(out, (s / np.float32(3)).reshape(1, 3))


def test_output_is_ordered_by_voxel_index_x_fastest():
    a = np.array([[1.2, 0.1, 0.1], [0.1, 0.1, 0.6], [0.1, 0.7, 0.1], [0.1, 0.1, 0.1]], np.float32)
    out = po.voxel_grid(a, 0.5)
    np.testing.assert_array_equal(out, a[[3, 0, 2, 1]])


def test_negative_coordinates_use_floor_relative_to_min():
    a = np.array([[-0.1, -0.1, -0.1], [0.1, 0.1, 0.1], [-0.6, -0.1, -0.1]], np.float32)
    out = po.voxel_grid(a, 0.5)
    np.testing.assert_array_equal(out, a[[2, 0, 1]])


def test_random_clouds_match_the_numpy_restatement():
    rng = np.random.default_rng(1)
    for n, leaf, span in ((5000, 0.5, 20.0), (20000, 0.3, 8.0), (3000, 1.0, 200.0)):
        a = rng.uniform(-span, span, (n, 3)).astype(np.float32)
        a[: n // 10] = a[n // 10: 2 * (n // 10)]          # exact duplicates
        out = po.voxel_grid(a, leaf)
        np.testing.assert_array_equal(out.view(np.uint32), ref_numpy(a, leaf).view(np.uint32))


def test_leaf_too_small_returns_the_input_unchanged():
    a = np.array([[0, 0, 0], [3000.0, 3000.0, 3000.0], [1.0, 2.0, 3.0]], np.float32)
    out = po.voxel_grid(a, 0.001)      # (3000/0.001)^3 voxels overflow int32
    np.testing.assert_array_equal(out, a)


def test_empty_cloud():
    assert len(po.voxel_grid(np.zeros((0, 3), np.float32), 0.5)) == 0
