"""Known-answer tests for the oracle's incremental-map restatement (oracle_path.c: orc_map_add /
orc_map_delete_boxes), the stand-in for ikd-Tree Add_Points / Delete_Point_Boxes (src/laserMapping.cpp:470-471,
:275).  ikd-Tree itself is an un-vendored dependency of the reference (include/ikd-Tree, KD_TREE::Add_Points);
these cases pin the published semantics: with down-sampling on, the point nearest to the voxel centre survives."""
import numpy as np

from oracle import pyoracle as po


def vox(p, ds):
    return tuple(np.floor(np.asarray(p, np.float64) / ds).astype(np.int64))


def test_add_without_downsample_appends_in_order():
    m = np.random.default_rng(0).uniform(-5, 5, (50, 3)).astype(np.float32)
    a = np.random.default_rng(1).uniform(-5, 5, (20, 3)).astype(np.float32)
    out = po.map_add(m, a, downsample=False)
    np.testing.assert_array_equal(out, np.vstack([m, a]))


def test_add_into_empty_voxel_keeps_nearest_to_centre_later_wins_ties():
    ds = 0.5
    m = np.zeros((0, 3), np.float32)
    # voxel (0,0,0): centre 0.25^3.  p1 at distance^2 3*0.01, p2 closer, p3 mirrors p2 (exact tie) -> p3 survives
    a = np.array([[0.15, 0.15, 0.15], [0.2, 0.2, 0.2], [0.3, 0.3, 0.3], [1.1, 0.1, 0.1]], np.float32)
    out = po.map_add(m, a, True, ds)
    d2 = ((np.float32(0.2) - np.float32(0.25)) ** 2, (np.float32(0.3) - np.float32(0.25)) ** 2)
    want_first = a[2] if d2[1] <= d2[0] else a[1]
    np.testing.assert_array_equal(out, np.vstack([want_first, a[3]]))


def test_single_existing_point_that_stays_nearest_is_untouched_and_new_point_dropped():
    ds = 0.5
    m = np.array([[0.26, 0.25, 0.25], [3.0, 3.0, 3.0]], np.float32)
    a = np.array([[0.05, 0.05, 0.05]], np.float32)
    out = po.map_add(m, a, True, ds)
    np.testing.assert_array_equal(out, m)


def test_new_point_displaces_existing_and_exact_tie_goes_to_new():
    ds = 0.5
    m = np.array([[0.125, 0.25, 0.25], [3.0, 3.0, 3.0]], np.float32)   # distance 0.125 from the centre
    a = np.array([[0.375, 0.25, 0.25]], np.float32)                    # the same distance, exactly
    out = po.map_add(m, a, True, ds)
    np.testing.assert_array_equal(out, np.array([[3.0, 3.0, 3.0], [0.375, 0.25, 0.25]], np.float32))


def test_crowded_voxel_collapses_to_its_best_point_when_touched():
    ds = 0.5
    m = np.array([[0.05, 0.05, 0.05], [0.24, 0.25, 0.25], [0.4, 0.4, 0.4], [0.24, 0.25, 0.25]], np.float32)
    a = np.array([[0.45, 0.45, 0.45]], np.float32)
    out = po.map_add(m, a, True, ds)
    # best existing stays (lowest index among the two identical points); everything else in the voxel goes
    np.testing.assert_array_equal(out, m[1:2])


def test_add_random_every_touched_voxel_holds_exactly_one_point():
    rng = np.random.default_rng(3)
    ds = 0.5
    m = po.map_add(np.zeros((0, 3), np.float32), rng.uniform(-4, 4, (3000, 3)).astype(np.float32), True, ds)
    a = rng.uniform(-4, 4, (2000, 3)).astype(np.float32)
    out = po.map_add(m, a, True, ds)
    touched = {vox(p, ds) for p in a}
    counts = {}
    for p in out:
        counts[vox(p, ds)] = counts.get(vox(p, ds), 0) + 1
    assert all(counts[v] == 1 for v in touched)
    # survivor of a voxel is the candidate nearest to the centre
    cand = {}
    for p in np.vstack([m, a]):
        cand.setdefault(vox(p, ds), []).append(p)
    for p in out:
        v = vox(p, ds)
        if v in touched:
            c = (np.array(v, np.float64) * ds + 0.5 * ds).astype(np.float32)
            dmin = min(float(((q - c) ** 2).sum()) for q in cand[v])
            assert float(((p - c) ** 2).sum()) <= dmin * (1 + 1e-5)
    # untouched voxels are carried over verbatim, in order
    keep = np.array([vox(p, ds) not in touched for p in m])
    keep_out = np.array([vox(p, ds) not in touched for p in out])
    np.testing.assert_array_equal(out[keep_out], m[keep])


def test_delete_boxes_half_open():
    m = np.array([[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [2, 2, 2]], np.float32)
    out = po.map_delete_boxes(m, np.array([[0, 0, 0, 1, 1, 1]], np.float32))
    np.testing.assert_array_equal(out, m[[1, 3]])
    out = po.map_delete_boxes(m, np.array([[0, 0, 0, 1, 1, 1], [1.5, 1.5, 1.5, 2.5, 2.5, 2.5]], np.float32))
    np.testing.assert_array_equal(out, m[[1]])
    np.testing.assert_array_equal(po.map_delete_boxes(m, np.zeros((0, 6), np.float32)), m)
