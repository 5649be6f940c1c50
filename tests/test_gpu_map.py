"""GPU parity for SURVEY.md 8(f) row 1: map_incremental (src/laserMapping.cpp:427-474) and the incremental map
(ikdtree.Add_Points / Delete_Point_Boxes) against the oracle's restatement.  Everything here is index / flag /
fp32-copy work, so the bar is bit-exact: same classes, same surviving points in the same order, and a search on
the updated map that matches an oracle kd-tree built from the oracle's updated map."""
import numpy as np
import pytest

from fast_lio_amd import capi, synth
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

DS = 0.5


def same_points(a, b, msg=""):
    assert a.shape == b.shape, f"{msg}: {a.shape} vs {b.shape}"
    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32), err_msg=msg)


@pytest.fixture(scope="module")
def prob():
    pr = synth.make_problem(150000, 12000, "avia", cfg=1)
    return pr


def search_matches(h, map_xyz, body, x):
    m = po.Map(map_xyz)
    sc = po.Scan(body, nthreads=8)
    h.scan_upload(body)
    HTH, HTh, n_eff, _ = h.eval(x, True, False)
    sc.h_share_model(m, x, True, False)
    np.testing.assert_array_equal(h.fetch_selected(), sc.selected)
    assert n_eff == sc.n_eff
    idx, d2, cnt = h.fetch_neighbors()
    gate = (sc.nn_cnt == 5) & (sc.nn_d2[:, 4] <= 5.0)
    np.testing.assert_array_equal(idx[gate], sc.nn_idx[gate])
    np.testing.assert_array_equal(d2[gate].view(np.uint32), sc.nn_d2[gate].view(np.uint32))
    sel = sc.selected.astype(bool)
    np.testing.assert_array_equal(h.fetch_normvec()[sel].view(np.uint32), sc.normvec[sel].view(np.uint32))
    return int(gate.sum())


def test_download_returns_the_built_map(prob):
    h = capi.Handle()
    h.map_build(prob.map_xyz)
    same_points(h.map_download(), prob.map_xyz.astype(np.float32))


@pytest.mark.parametrize("downsample", [True, False])
def test_map_add_matches_oracle_and_search_stays_exact(prob, downsample):
    pr = prob
    rng = np.random.default_rng(11)
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    # new points: jittered copies of map points (crowded voxels), exact duplicates, a block outside the old AABB
    pick = rng.integers(0, len(pr.map_xyz), 15000)
    jit = pr.map_xyz[pick] + rng.normal(0, 0.2, (15000, 3)).astype(np.float32)
    dup = pr.map_xyz[rng.integers(0, len(pr.map_xyz), 500)]
    far = pr.map_xyz.max(0) + rng.uniform(1.0, 30.0, (800, 3)).astype(np.float32)
    add = np.vstack([jit, dup, far, jit[:300]]).astype(np.float32)
    want = po.map_add(pr.map_xyz, add, downsample, DS)
    h.map_add(add, downsample, DS)
    assert h.M == len(want)
    got = h.map_download()
    same_points(got, want, "map after Add_Points")
    assert search_matches(h, want, pr.body, pr.x_true) > 1000
    # a second round on top of the first
    add2 = (want[rng.integers(0, len(want), 6000)] + rng.normal(0, 0.1, (6000, 3))).astype(np.float32)
    want2 = po.map_add(want, add2, downsample, DS)
    h.map_add(add2, downsample, DS)
    same_points(h.map_download(), want2, "second Add_Points")


def test_map_add_exact_ties_on_a_lattice():
    # every coordinate a multiple of 1/8: distances to the voxel centres tie exactly, all the time
    rng = np.random.default_rng(5)
    m = (rng.integers(-40, 40, (4000, 3)) / 8.0).astype(np.float32)
    add = (rng.integers(-40, 40, (3000, 3)) / 8.0).astype(np.float32)
    h = capi.Handle()
    h.map_build(m)
    want = po.map_add(m, add, True, DS)
    h.map_add(add, True, DS)
    same_points(h.map_download(), want, "lattice ties")
    # and with another voxel size that does not divide the lattice
    want2 = po.map_add(want, add[::-1].copy(), True, 0.3)
    h.map_add(add[::-1].copy(), True, 0.3)
    same_points(h.map_download(), want2, "lattice ties, ds=0.3")


def test_map_add_into_empty_and_unbuilt_maps():
    rng = np.random.default_rng(2)
    add = rng.uniform(-3, 3, (2000, 3)).astype(np.float32)
    h = capi.Handle()           # never built
    h.map_add(add, True, DS)
    same_points(h.map_download(), po.map_add(np.zeros((0, 3), np.float32), add, True, DS), "unbuilt")
    h2 = capi.Handle()
    h2.map_build(np.zeros((0, 3), np.float32))
    h2.map_add(add, False, DS)
    same_points(h2.map_download(), add, "empty, no down-sampling")
    h2.map_add(np.zeros((0, 3), np.float32), True, DS)   # adding nothing changes nothing
    same_points(h2.map_download(), add)


def test_delete_point_boxes(prob):
    pr = prob
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    c = np.median(pr.map_xyz, axis=0)
    boxes = np.array([np.r_[c - 6, c + 6], np.r_[c + [10, -3, -50], c + [25, 9, 50]]], np.float32)
    want = po.map_delete_boxes(pr.map_xyz, boxes)
    assert 0 < len(want) < len(pr.map_xyz)
    h.map_delete_boxes(boxes)
    same_points(h.map_download(), want, "Delete_Point_Boxes")
    search_matches(h, want, pr.body, pr.x_true)
    # delete everything
    big = np.array([[-1e6, -1e6, -1e6, 1e6, 1e6, 1e6]], np.float32)
    h.map_delete_boxes(big)
    assert h.M == 0
    h.scan_upload(pr.body[:100])
    assert h.eval(pr.x_true, True, False)[2] == 0


@pytest.mark.parametrize("fsm", [0.5, 0.3])
def test_map_incremental_matches_oracle(prob, fsm):
    pr = prob
    m = po.Map(pr.map_xyz)
    xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    # scan with a tail of points nowhere near the map (no neighbour inside the search bound)
    rng = np.random.default_rng(3)
    tail = (pr.body[:400] * 0.0 + rng.uniform(150, 400, (400, 3))).astype(np.float32)
    # ... and of points in the empty space in and around the map, metres to tens of metres from the nearest map point (their
    # points_near[0] comes from the shell search of k_far_search, which must stop at exactly the right shell)
    lo, hi = pr.map_xyz.min(0) - 30.0, pr.map_xyz.max(0) + 30.0
    Rw, tw = synth.quat_to_R(pr.x_true[3:7]), pr.x_true[0:3]
    gaps_world = rng.uniform(lo, hi, (1500, 3))
    gaps = ((gaps_world - tw) @ Rw).astype(np.float32)   # body frame of the true pose (identity extrinsic in this problem or not: any body point will do)
    body = np.vstack([pr.body, tail, gaps]).astype(np.float32)
    h.scan_upload(body)
    sc = po.Scan(body, nthreads=8)
    # the passes of one update, driven identically on both sides: search at the prior, two re-linearisations
    x2 = po.state_boxplus(xp, np.r_[0.02, -0.01, 0.015, 0.002, -0.001, 0.003, np.zeros(17)])
    for x, s in ((xp, True), (x2, False), (pr.x_true, False)):
        h.eval(x, s, False)
        sc.h_share_model(m, x, s, False)
    x_post = pr.x_true
    w_ref, c_ref = sc.map_incremental_classify(m, x_post, fsm, True)
    n1, n2 = h.map_incremental(x_post, fsm, True, apply=False)
    w, c = h.fetch_map_incremental()
    same_points(w, w_ref, "feats_down_world")
    np.testing.assert_array_equal(c, c_ref, err_msg="add / no-downsample / skip classes")
    assert (n1, n2) == (int((c_ref == 1).sum()), int((c_ref == 2).sum()))
    assert n1 > 0 and n2 > 0 and int((c_ref == 0).sum()) > 0
    # apply: Add_Points(PointToAdd, true); Add_Points(PointNoNeedDownsample, false)
    want = po.map_add(pr.map_xyz, w_ref[c_ref == 1], True, fsm)
    want = po.map_add(want, w_ref[c_ref == 2], False, fsm)
    assert h.map_incremental(x_post, fsm, True, apply=True) == (n1, n2)
    same_points(h.map_download(), want, "map after map_incremental")
    # the neighbour cache died with the old map
    with pytest.raises(capi.FlhError):
        h.eval(x_post, False, False)
    assert search_matches(h, want, pr.body, pr.x_true) > 1000


def test_map_incremental_before_ekf_init_adds_everything(prob):
    pr = prob
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    h.scan_upload(pr.body)
    h.eval(pr.x_true, True, False)
    n1, n2 = h.map_incremental(pr.x_true, DS, False, apply=False)
    assert (n1, n2) == (len(pr.body), 0)
    sc = po.Scan(pr.body, nthreads=8)
    m = po.Map(pr.map_xyz)
    sc.h_share_model(m, pr.x_true, True, False)
    w_ref, c_ref = sc.map_incremental_classify(m, pr.x_true, DS, False)
    w, c = h.fetch_map_incremental()
    np.testing.assert_array_equal(c, c_ref)
    same_points(w, w_ref)


def test_map_incremental_tiny_maps(prob):
    # fewer than NUM_MATCH_POINTS map points: points_near.size() < 5 -> the veto loop breaks at once (:454)
    pr = prob
    for M in (1, 3, 4, 5, 7):
        mp = np.ascontiguousarray(pr.map_xyz[:M])
        m = po.Map(mp)
        h = capi.Handle()
        h.map_build(mp)
        h.scan_upload(pr.body[:3000])
        sc = po.Scan(pr.body[:3000], nthreads=4)
        h.eval(pr.x_true, True, False)
        sc.h_share_model(m, pr.x_true, True, False)
        w_ref, c_ref = sc.map_incremental_classify(m, pr.x_true, DS, True)
        h.map_incremental(pr.x_true, DS, True, apply=False)
        w, c = h.fetch_map_incremental()
        np.testing.assert_array_equal(c, c_ref, err_msg=f"M={M}")


def test_full_scan_cycle_update_then_map_incremental(prob):
    """One laserMapping loop body (:879-927): iterated update, then map_incremental with the posterior state, then
    the next scan's update on the grown map -- GPU and oracle side by side."""
    pr = prob
    m = po.Map(pr.map_xyz)
    xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    h.scan_upload(pr.body)
    kf = capi.Esekf(h, max_iter=3, extrinsic_est_en=False)
    kf.change_x(xp)
    kf.change_P(P)
    kf.update(0.001)
    x_post = kf.get_x()
    sc = po.Scan(pr.body, nthreads=8)
    sc.update_iterated(m, xp, P)
    w_ref, c_ref = sc.map_incremental_classify(m, x_post, DS, True)
    n1, n2 = h.map_incremental(x_post, DS, True, apply=True)
    w, c = h.fetch_map_incremental()
    np.testing.assert_array_equal(c, c_ref)
    same_points(w, w_ref)
    want = po.map_add(po.map_add(pr.map_xyz, w_ref[c_ref == 1], True, DS), w_ref[c_ref == 2], False, DS)
    same_points(h.map_download(), want)
    # next scan against the grown map
    pr2 = synth.make_problem(150000, 12000, "avia", cfg=1, scan_seed=7)
    m2 = po.Map(want)
    h.scan_upload(pr2.body)
    xp2, P2 = synth.propagate_prior_cov(capi.predict_fn, pr2.x_prior)
    kf.change_x(xp2)
    kf.change_P(P2)
    st = kf.update(0.001)
    sc2 = po.Scan(pr2.body, nthreads=8)
    x_ref, P_ref, st_ref = sc2.update_iterated(m2, xp2, P2)
    assert list(st.n_eff)[: st.passes] == list(st_ref.n_eff)[: st.passes]
    assert np.linalg.norm(kf.get_x()[0:3] - x_ref[0:3]) <= 1e-4
    np.testing.assert_array_equal(h.fetch_selected(), sc2.selected)


def test_map_incremental_requires_a_searched_scan(prob):
    h = capi.Handle()
    h.map_build(prob.map_xyz[:1000])
    with pytest.raises(capi.FlhError):
        h.map_incremental(prob.x_true, DS)
    h.scan_upload(prob.body[:100])
    with pytest.raises(capi.FlhError):
        h.map_incremental(prob.x_true, DS)
    with pytest.raises(capi.FlhError):
        h.map_add(np.array([[np.nan, 0, 0]], np.float32))


def test_fov_segment_moves_cube_and_deletes_slabs(prob):
    """lasermap_fov_segment (:230-280) through the C ABI: same slabs as the oracle, and the slabs are gone from the map."""
    pr = prob
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    c = np.median(pr.map_xyz, axis=0).astype(np.float64)
    lm_g, lm_o = capi.FlhLocalMap(), po.LocalMap()
    cur = pr.map_xyz.astype(np.float32)
    ext = float(np.ptp(pr.map_xyz, axis=0).max())
    cube, det = 0.8 * ext, 0.1 * ext
    path = [c, c + [0.2 * ext, 0, 0], c + [0.3 * ext, 0.05 * ext, 0], c + [0.3 * ext, -0.3 * ext, 0], c]
    total = 0
    for p in path:
        want = po.fov_segment(lm_o, p, cube, det)
        boxes, ndel = h.fov_segment(lm_g, p, cube, det)
        np.testing.assert_array_equal(boxes.view(np.uint32), want.view(np.uint32))
        assert list(lm_g.vertex_min) == list(lm_o.vertex_min) and list(lm_g.vertex_max) == list(lm_o.vertex_max)
        new = po.map_delete_boxes(cur, want) if len(want) else cur
        assert ndel == len(cur) - len(new)
        cur = new
        same_points(h.map_download(), cur, "map after lasermap_fov_segment")
        total += ndel
    assert total > 0


def test_long_stream_of_removals_keeps_what_the_search_reads_bounded(prob):
    """lasermap_fov_segment -> Delete_Point_Boxes runs for the life of the node (src/laserMapping.cpp:231-277): 200 steps of a
    sensor wandering through the map, the local-map cube following it (slabs removed whenever it moves) and the scan's points going
    back in around the sensor (Add_Points with down-sampling).  A removal tombstones slots in place; bricks that only lose points
    are compacted once their live points fall below half of their range (k_brick_purge), so the slots inside the bricks' ranges --
    what every later search of those cells loads -- stay below 2 x the live points (+ 8 per brick: ranges too short to bother with); the map
    stays the oracle's point for point, and a search on it at the end is still exact."""
    pr = prob
    rng = np.random.default_rng(77)
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    orig = pr.map_xyz.astype(np.float32)
    lo, hi = orig.min(0).astype(np.float64), orig.max(0).astype(np.float64)
    ext = float((hi - lo).max())
    cube, det = 0.9 * ext, 0.1 * ext
    lm_g, lm_o = capi.FlhLocalMap(), po.LocalMap()
    cur = orig
    pos = np.median(orig, axis=0).astype(np.float64)
    vel = np.array([0.035 * ext, 0.02 * ext, 0.0])
    removed = added = moves = 0
    worst = 0.0
    for step in range(200):
        pos = pos + vel
        for d in range(2):  # bounce off the map's walls
            if pos[d] < lo[d] + 0.1 * ext or pos[d] > hi[d] - 0.1 * ext:
                vel[d] = -vel[d]
        want = po.fov_segment(lm_o, pos, cube, det)
        boxes, ndel = h.fov_segment(lm_g, pos, cube, det)
        np.testing.assert_array_equal(boxes.view(np.uint32), want.view(np.uint32))
        if len(want):
            new = po.map_delete_boxes(cur, want)
            assert ndel == len(cur) - len(new)
            removed += ndel
            moves += 1
            cur = new
        # the scan's points return around the sensor: jittered copies of the original map's points within reach
        near = orig[np.linalg.norm(orig - pos.astype(np.float32), axis=1) < 0.12 * ext]
        if len(near):
            add = (near[rng.integers(0, len(near), 400)] + rng.normal(0, 0.05, (400, 3))).astype(np.float32)
            before = len(cur)
            cur = po.map_add(cur, add, True, DS)
            h.map_add(add, True, DS)
            added += len(cur) - before
        st = h.map_storage_stats()
        assert st["points"] == len(cur)
        assert st["slots_in_brick_ranges"] <= 2 * st["points"] + 8 * st["bricks"], (step, st)
        worst = max(worst, st["slots_in_brick_ranges"] / max(st["points"], 1))
        if step % 25 == 24 or step == 199:
            same_points(h.map_download(), cur, f"map after step {step}")
    st = h.map_storage_stats()
    ms = h.map_stats()
    print(f"[200-step removal stream] cube moved {moves}x, {removed} points removed, {added} added, {len(cur)} left of {len(orig)}; "
          f"bricks purged {st['bricks_purged']}, slots in brick ranges / points: worst {worst:.2f}, now "
          f"{st['slots_in_brick_ranges'] / max(st['points'], 1):.2f}; re-indexings {ms['reindex']}, brick-wise changes {ms['brickwise']}")
    assert moves >= 5 and removed > 10000 and st["bricks_purged"] > 0
    assert search_matches(h, cur, pr.body, pr.x_true) >= 0
    h.close()


# ---------------------------------------------------------------------------------------------- 8(f) row 2
def raw_scan(pr, n, seed):
    """An un-down-sampled scan: the down-sampled one plus dense clutter around its points (several per leaf)."""
    rng = np.random.default_rng(seed)
    base = pr.body[rng.integers(0, len(pr.body), n)]
    return (base + rng.normal(0, 0.15, (n, 3))).astype(np.float32)


@pytest.mark.parametrize("leaf", [0.5, 0.2])
def test_scan_voxel_grid_matches_oracle_bit_for_bit(prob, leaf):
    pr = prob
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    raw = raw_scan(pr, 60000, 1)
    raw[:500] = raw[500:1000]                       # exact duplicates
    want = po.voxel_grid(raw, leaf)
    n = h.scan_stage_downsampled(2, raw, leaf)
    assert n == len(want) and 0 < n < len(raw)
    h.scan_activate(2)
    assert h.N == n
    same_points(h.fetch_scan(), want, "feats_down_body")
    # the staged scan behaves exactly like the same points handed to flh_scan_upload
    a = h.eval(pr.x_true, True, False)
    sel_a = h.fetch_selected()
    h.scan_upload(want)
    b = h.eval(pr.x_true, True, False)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(sel_a, h.fetch_selected())
    assert a[2] == b[2] and a[2] > 100


def test_scan_voxel_grid_edge_cases(prob):
    h = capi.Handle()
    h.map_build(prob.map_xyz[:1000])
    assert h.scan_stage_downsampled(0, np.zeros((0, 3), np.float32), 0.5) == 0
    one = np.array([[1.0, 2.0, 3.0]], np.float32)
    assert h.scan_stage_downsampled(0, one, 0.5) == 1
    h.scan_activate(0)
    same_points(h.fetch_scan(), one)
    far = np.array([[0, 0, 0], [3000.0, 3000.0, 3000.0], [1.0, 2.0, 3.0]], np.float32)
    assert h.scan_stage_downsampled(1, far, 0.001) == 3           # leaf too small: input returned unchanged
    h.scan_activate(1)
    same_points(h.fetch_scan(), far)
    with pytest.raises(capi.FlhError):
        h.scan_stage_downsampled(0, np.array([[np.inf, 0, 0]], np.float32), 0.5)
    with pytest.raises(capi.FlhError):
        h.scan_stage_downsampled(0, one, 0.0)
    # a voxel with thousands of points (dense clutter next to the sensor)
    rng = np.random.default_rng(9)
    blob = np.vstack([rng.uniform(0.0, 0.49, (5000, 3)), rng.uniform(-30, 30, (2000, 3))]).astype(np.float32)
    n = h.scan_stage_downsampled(3, blob, 0.5)
    h.scan_activate(3)
    same_points(h.fetch_scan(), po.voxel_grid(blob, 0.5))
    assert n == h.N


# ---------------------------------------------------------------------------------------------- 8(f) row 3
def imu_poses_for_test(x0, n_imu=21, T=0.1, seed=4):
    """IMUpose as UndistortPcl's forward half builds it (IMU_Processing.hpp:240-300): one predict per IMU sample."""
    rng = np.random.default_rng(seed)
    x = np.array(x0, np.float64)
    P = po.init_P()
    Q = po.process_noise_cov()
    rows = []
    dt = T / (n_imu - 1)
    gyro0, acc0 = np.array([0.4, -0.3, 0.9]), np.array([0.8, -0.5, 9.9])

    def rotm(q):
        return np.array([po.quat_rot(q, e) for e in np.eye(3)]).T

    rows.append((0.0, (0, 0, 0), (0, 0, 0), x[14:17], x[0:3], rotm(x[3:7]).reshape(9)))
    for k in range(1, n_imu):
        gyro = gyro0 + rng.normal(0, 0.05, 3)
        acc = acc0 + rng.normal(0, 0.2, 3)
        x, P = po.predict(x, P, dt, Q, acc, gyro)
        R = rotm(x[3:7])
        acc_s = R @ (acc - x[20:23]) + x[23:26]
        rows.append((k * dt, acc_s, gyro - x[17:20], x[14:17], x[0:3], R.reshape(9)))
    return po.make_poses(rows), x


def test_undistortion_matches_oracle(prob):
    pr = prob
    h = capi.Handle()
    h.map_build(pr.map_xyz[:1000])
    x0 = pr.x_true.copy()
    x0[14:17] = (8.0, -3.0, 0.5)                       # moving fast enough for the correction to matter
    poses, x_end = imu_poses_for_test(x0)
    rng = np.random.default_rng(6)
    raw = raw_scan(pr, 50000, 2)
    tms = rng.uniform(-2.0, 104.0, len(raw)).astype(np.float32)   # a few before the first / after the last IMU pose
    tms[:50] = 0.0
    tms[50:100] = np.float32(50.0)                                # exactly on an IMU sample
    pts = np.c_[raw, tms].astype(np.float32)
    want = po.undistort(poses, x_end, pts)
    n, und = h.scan_stage_undistorted(1, pts, poses, x_end, leaf_size=0.0)
    assert n == len(pts)
    moved = np.linalg.norm(want - raw, axis=1)
    assert moved.max() > 0.2 and (moved[tms <= 0] == 0).all()
    # double arithmetic in the same order on both sides; sin/cos come from different libms, so a result may differ in
    # its last float bit: bound the error at float rounding of the coordinates and require near-total bit equality
    scale = np.abs(want).max()
    assert np.abs(und - want).max() <= 2.0 * np.spacing(np.float32(scale))
    assert (und.view(np.uint32) == want.view(np.uint32)).mean() > 0.999
    h.scan_activate(1)
    same_points(h.fetch_scan(), und, "leaf_size <= 0 stages feats_undistort as is")
    # the chain undistort -> voxel grid -> staging
    n2, und2 = h.scan_stage_undistorted(2, pts, poses, x_end, leaf_size=0.5)
    same_points(und2, und)
    h.scan_activate(2)
    same_points(h.fetch_scan(), po.voxel_grid(und, 0.5), "feats_down_body from the device chain")
    assert n2 == h.N
    # the first IMU sample older than the first point: IMUpose[1].offset_time < IMUpose[0].offset_time = 0 (a normal case)
    poses[1].offset_time = -0.002
    want3 = po.undistort(poses, x_end, pts)
    _, und3 = h.scan_stage_undistorted(0, pts, poses, x_end, leaf_size=0.0)
    assert np.abs(und3 - want3).max() <= 2.0 * np.spacing(np.float32(scale))
    assert (und3.view(np.uint32) == want3.view(np.uint32)).mean() > 0.999
    assert np.abs(want3 - want).max() > 0      # the early points did change segment
    with pytest.raises(capi.FlhError):
        poses[3].offset_time = float("nan")
        h.scan_stage_undistorted(0, pts[:10], poses, x_end)


def test_undistortion_earliest_point_younger_than_the_second_imu_pose(prob):
    """src/IMU_Processing.hpp:345: the reference's sweep compensates the earliest point of the cloud once per segment older than
    it.  Every point of this cloud is younger than IMUpose[3], so the earliest one is carried four times; the device must do
    the same (default) or carry it once (flh_config.undistort_first_point = 0), as the oracle does either way."""
    pr = prob
    x0 = pr.x_true.copy()
    x0[14:17] = (8.0, -3.0, 0.5)
    poses, x_end = imu_poses_for_test(x0)
    rng = np.random.default_rng(16)
    raw = raw_scan(pr, 20000, 2)
    t3 = 1000.0 * poses[3].offset_time
    tms = rng.uniform(t3 + 0.5, 104.0, len(raw)).astype(np.float32)
    first = 1234
    tms[first] = np.float32(t3 + 0.25)
    tms[first + 4000] = tms[first]           # an equal time at a higher index: not "the first point"
    pts = np.c_[raw, tms].astype(np.float32)
    want = po.undistort(poses, x_end, pts)
    once = po.undistort(poses, x_end, pts, first_point=False)
    assert (np.abs(want - once).max(axis=1) > 0).sum() == 1 and np.abs(want[first] - once[first]).max() > 1e-3
    scale = np.abs(want).max()
    for flag, ref in ((-1, want), (0, once)):
        h = capi.Handle(undistort_first_point=flag)
        h.map_build(pr.map_xyz[:1000])
        n, und = h.scan_stage_undistorted(1, pts, poses, x_end, leaf_size=0.0)
        assert n == len(pts)
        assert np.abs(und - ref).max() <= 2.0 * np.spacing(np.float32(scale))
        assert (und.view(np.uint32) == ref.view(np.uint32)).mean() > 0.999
        assert np.abs(und[first] - ref[first]).max() <= 4.0 * np.spacing(np.float32(scale))  # four passes, one ulp each at most
        h.close()


def test_velodyne_stream_with_incremental_map():
    """BASELINE configs[2] in miniature: a Velodyne-16 scan stream whose points are inserted into the map after every
    update (laserMapping.cpp:879-927 loop body, five times over): the device map must stay bit-identical to the
    oracle's, so that every later search, flag and posterior keeps matching."""
    M, N = 250000, 9000
    pr0 = synth.make_problem(M, N, "velodyne", cfg=3)
    scene = pr0.scene
    h = capi.Handle()
    h.map_build(pr0.map_xyz)
    kf = capi.Esekf(h, max_iter=3, extrinsic_est_en=False)
    cur = pr0.map_xyz.astype(np.float32)
    lm_g, lm_o = capi.FlhLocalMap(), po.LocalMap()
    ext = float(np.ptp(cur, axis=0).max())
    grown = 0
    for k in range(5):
        pr = synth.make_problem(M, N, "velodyne", cfg=3, scan_seed=k, scene=scene)
        xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
        # lasermap_fov_segment first (:886), with a cube small enough to bite
        want_boxes = po.fov_segment(lm_o, xp[0:3], 0.9 * ext, 0.12 * ext)
        boxes, ndel = h.fov_segment(lm_g, xp[0:3], 0.9 * ext, 0.12 * ext)
        np.testing.assert_array_equal(boxes.view(np.uint32), want_boxes.view(np.uint32))
        if len(want_boxes):
            cur = po.map_delete_boxes(cur, want_boxes)
        m = po.Map(cur)
        h.scan_upload(pr.body)
        st = kf.update_scan(-1, np.ascontiguousarray(xp), np.ascontiguousarray(P), 0.001)
        sc = po.Scan(pr.body, nthreads=8)
        x_ref, P_ref, st_ref = sc.update_iterated(m, xp, P)
        assert list(st.n_eff)[: st.passes] == list(st_ref.n_eff)[: st.passes], f"scan {k}"
        np.testing.assert_array_equal(h.fetch_selected(), sc.selected, err_msg=f"scan {k}")
        x_post = kf.get_x()
        assert np.linalg.norm(x_post[0:3] - x_ref[0:3]) <= 1e-4
        w_ref, c_ref = sc.map_incremental_classify(m, x_post, DS, True)
        n1, n2 = h.map_incremental(x_post, DS, True, apply=True)
        w, c = h.fetch_map_incremental()
        np.testing.assert_array_equal(c, c_ref, err_msg=f"scan {k}")
        new = po.map_add(po.map_add(cur, w_ref[c_ref == 1], True, DS), w_ref[c_ref == 2], False, DS)
        grown += len(new) - len(cur)
        cur = new
        same_points(h.map_download(), cur, f"map after scan {k}")
    assert grown > 0


def test_map_incremental_enqueued_without_the_hosts_wait():
    """The node's loop never asks map_incremental for its two list lengths; from the second small change on, Add_Points is then
    enqueued right behind the classification with the lengths read on the device (flh_map_change_stats counts them).  Same map
    as the oracle's after every scan -- and also after a change that outgrows its launches (a scan of 12 000 points over new
    ground after a small one: more than 8 192 to insert), which the library replays when it collects the change's counters.  The
    launches are sized from the previous change: a second large change right after a large one goes through without the wait and
    without a replay, on the general path (scan + device-wide sort over the bound, the lengths still read on the device)."""
    M, N = 250000, 9000
    pr0 = synth.make_problem(M, N, "velodyne", cfg=3)
    scene = pr0.scene
    h = capi.Handle()
    h.map_build(pr0.map_xyz)
    cur = pr0.map_xyz.astype(np.float32)
    rng = np.random.default_rng(5)
    large = (3, 5, 6)
    for k in range(7):
        pr = synth.make_problem(M, N, "velodyne", cfg=3, scan_seed=10 + k, scene=scene)
        body = np.ascontiguousarray(pr.body[:4000])  # at most 4 000 points to insert: a change of a scan's usual size
        if k in large:  # new ground: points in the empty space above the scene, all of them inserted
            far = rng.uniform(-60, 60, (12000, 3)).astype(np.float32)
            far[:, 2] = rng.uniform(200 + 100 * k, 260 + 100 * k, 12000).astype(np.float32)
            body = np.ascontiguousarray(far)
        m = po.Map(cur)
        h.scan_upload(body)
        h.eval(pr.x_true, True, False)
        sc = po.Scan(body, nthreads=8)
        sc.h_share_model(m, pr.x_true, True, False)
        w_ref, c_ref = sc.map_incremental_classify(m, pr.x_true, DS, True)
        assert h.map_incremental(pr.x_true, DS, True, apply=True, counts=(k == 0)) is None or k == 0
        assert int((c_ref != 0).sum()) > (8192 if k in large else 0)
        cur = po.map_add(po.map_add(cur, w_ref[c_ref == 1], True, DS), w_ref[c_ref == 2], False, DS)
        same_points(h.map_download(), cur, f"map after scan {k}")
        assert search_matches(h, cur, pr.body[:3000], pr.x_true) > 500
    st = h.map_change_stats()
    # k = 3 and k = 5 follow a small change (launches for 8 192 points: replayed); k = 6 follows a large one (launches for 19 000)
    assert st["enqueued_without_wait"] >= 6 and st["replayed"] == 2, st
    h.close()


def test_first_search_enqueued_behind_the_previous_map_change():
    """A scan's first searching pass goes onto the stream BEHIND the previous scan's map change, before the host has seen the
    change's counters (flh_eval_begin; they are folded in flh_eval_end).  Every search must equal the oracle's on the oracle's
    map after the change -- also when the counters then say "not applied, replay" (a change that outgrew its launches) or "re-index"
    (points outside the grid): the library repeats the pass on the settled map and counts it."""
    M, N = 250000, 9000
    pr0 = synth.make_problem(M, N, "velodyne", cfg=3)
    scene = pr0.scene
    h = capi.Handle()
    h.map_build(pr0.map_xyz)
    cur = pr0.map_xyz.astype(np.float32)
    rng = np.random.default_rng(15)
    grown, outside = (3,), (5,)
    reindex0 = h.map_stats()["reindex"]
    for k in range(8):
        pr = synth.make_problem(M, N, "velodyne", cfg=3, scan_seed=40 + k, scene=scene)
        body = np.ascontiguousarray(pr.body[:4000])
        if k in grown:  # new ground: 12 000 points to insert behind a small change (launches for 8 192: replayed)
            far = rng.uniform(-60, 60, (12000, 3)).astype(np.float32)
            far[:, 2] = rng.uniform(300, 360, 12000).astype(np.float32)
            body = np.ascontiguousarray(far)
        if k in outside:  # a few points beyond the grid's extent: the change flags a re-index
            out = (cur.max(0) + rng.uniform(200.0, 260.0, (50, 3))).astype(np.float32)
            body = np.ascontiguousarray(np.vstack([body[:3000], out]))
        # (from k = 1 on the previous change is still under way here: nothing between it and this search asks for the map)
        assert search_matches(h, cur, body, pr.x_true) >= 0
        m = po.Map(cur)
        sc = po.Scan(body, nthreads=8)
        sc.h_share_model(m, pr.x_true, True, False)
        w_ref, c_ref = sc.map_incremental_classify(m, pr.x_true, DS, True)
        h.map_incremental(pr.x_true, DS, True, apply=True, counts=(k == 0))
        cur = po.map_add(po.map_add(cur, w_ref[c_ref == 1], True, DS), w_ref[c_ref == 2], False, DS)
    redone = h.search_redone()
    same_points(h.map_download(), cur, "map after the stream")
    st = h.map_change_stats()
    assert st["enqueued_without_wait"] >= 6 and st["replayed"] >= 1, st
    assert h.map_stats()["reindex"] > reindex0
    assert redone >= 2, redone   # the search behind the replayed change and the one behind the re-indexing change
    h.close()


# ---------------------------------------------------------------------------------------------- brick storage paths
def test_brickwise_updates_and_every_fallback(prob):
    """The map index is changed brick by brick; whatever does not fit falls back to a full re-indexing from the id-ordered
    array.  Each path must leave the same map (bit-exact vs the oracle) and a search that still matches."""
    pr = prob
    rng = np.random.default_rng(21)
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    cur = pr.map_xyz.astype(np.float32)
    s0 = h.map_stats()
    assert s0["reindex"] == 1 and s0["brickwise"] == 0

    def step(add, downsample, tag, expect_reindex):
        nonlocal cur
        before = h.map_stats()
        cur = po.map_add(cur, add, downsample, DS)
        h.map_add(add, downsample, DS)
        after = h.map_stats()
        assert (after["reindex"] - before["reindex"] == 1) == expect_reindex, (tag, before, after)
        assert h.M == len(cur), tag
        same_points(h.map_download(), cur, tag)
        return after

    # 1. ordinary insert with down-sampling: brick-wise, in place
    add = (cur[rng.integers(0, len(cur), 8000)] + rng.normal(0, 0.2, (8000, 3))).astype(np.float32)
    a = step(add, True, "in place", False)
    assert a["brickwise"] == 1
    # 2. bricks that outgrow their slack are relocated (storage top moves), new bricks appear next to the map
    centres = cur[rng.integers(0, len(cur), 40)]
    crowd = (np.repeat(centres, 80, axis=0) + rng.uniform(-1.0, 1.0, (3200, 3))).astype(np.float32)   # +80 points in ~40 bricks
    b = step(crowd, False, "relocation", False)
    assert b["slots_used"] > a["slots_used"]
    near = (cur.max(0) + rng.uniform(2.0, 25.0, (3000, 3))).astype(np.float32)     # inside the grid's padding
    c = step(near, True, "new bricks", False)
    assert c["bricks"] > b["bricks"]
    search_matches(h, cur, pr.body, pr.x_true)
    # 3. a point outside the grid forces a re-indexing (ids renumbered)
    far = (cur.max(0) + rng.uniform(200.0, 260.0, (50, 3))).astype(np.float32)
    d = step(far, True, "outside the grid", True)
    assert d["ids"] == h.M
    # 4. more points in one brick than the rewrite tile holds
    blob = (cur[123] + rng.uniform(-0.7, 0.7, (2600, 3))).astype(np.float32)
    step(blob, False, "brick beyond the LDS tile", True)
    # 5. the storage runs out of slack after enough relocations
    grew = False
    for k in range(12):
        heap = (cur[rng.integers(0, len(cur), 40000)] + rng.normal(0, 0.3, (40000, 3))).astype(np.float32)
        before = h.map_stats()
        cur = po.map_add(cur, heap, False, DS)
        h.map_add(heap, False, DS)
        grew = grew or h.map_stats()["reindex"] > before["reindex"]
        if grew:
            break
    assert grew, "storage never filled up"
    same_points(h.map_download(), cur, "after the storage filled up")
    # 6. deletions are tombstones: ids stay, positions reported to the caller are those of the downloaded array
    c0 = np.median(cur, axis=0)
    boxes = np.array([np.r_[c0 - 8, c0 + 8]], np.float32)
    cur = po.map_delete_boxes(cur, boxes)
    h.map_delete_boxes(boxes)
    st = h.map_stats()
    assert st["ids"] > h.M == len(cur)
    same_points(h.map_download(), cur, "after Delete_Point_Boxes")
    assert search_matches(h, cur, pr.body, pr.x_true) > 1000


# ---------------------------------------------------------------------------------------------- small map changes
@pytest.mark.parametrize("downsample", [True, False])
def test_small_changes_in_one_workgroup_equal_the_general_path(prob, downsample):
    """flh_config.fused_small_changes: a change of at most 8192 points gives its surviving points their ids and sorts them by
    brick in one workgroup; larger ones (and every one with the option off) take the scan + device-wide sort.  Same map, same
    bookkeeping, at the size limit and on both sides of it -- and of the sizes at which the workgroup changes the items a thread
    sorts (2048, 4096)."""
    pr = prob
    rng = np.random.default_rng(77)
    hs = [capi.Handle(fused_small_changes=f) for f in (1, 0)]
    for h in hs:
        h.map_build(pr.map_xyz)
    cur = pr.map_xyz.astype(np.float32)
    for n in (1, 2, 63, 1000, 2048, 2049, 4096, 4097, 8191, 8192, 8193, 5000):
        add = (cur[rng.integers(0, len(cur), n)] + rng.normal(0, 0.25, (n, 3))).astype(np.float32)
        if n >= 63:  # crowded voxels, exact duplicates, a few new bricks beside the map
            add[: n // 8] = add[n // 8: 2 * (n // 8)]
            add[-(n // 16):] = (cur.max(0) + rng.uniform(2.0, 20.0, (n // 16, 3))).astype(np.float32)
        cur = po.map_add(cur, add, downsample, DS)
        stats = []
        for h in hs:
            h.map_add(add, downsample, DS)
            same_points(h.map_download(), cur, f"n={n}")
            stats.append(h.map_stats())
        assert stats[0] == stats[1], (n, stats)
    assert search_matches(hs[0], cur, pr.body, pr.x_true) > 1000


def test_small_change_in_which_no_point_survives(prob):
    """Every new point loses its voxel to the point the map already holds there: nothing is inserted, no brick is rewritten --
    k_brick_rewrite_heads has no workgroup with a brick to take the ticket, its first workgroup publishes the change's counters.
    Then a change in which exactly one point survives (one workgroup, one ticket per level)."""
    pr = prob
    h = capi.Handle()
    cur = po.map_add(np.zeros((0, 3), np.float32), pr.map_xyz[:20000].astype(np.float32), True, DS)   # one point per voxel
    h.map_build(cur)
    centre = (np.floor(cur.astype(np.float64) / DS) * DS + 0.5 * DS).astype(np.float32)
    lose = cur[:3000] + (cur[:3000] - centre[:3000]) * np.float32(0.05)                                # a little farther from the centre
    same_voxel = (np.floor(lose.astype(np.float64) / DS) == np.floor(cur[:3000].astype(np.float64) / DS)).all(1)
    lose = np.ascontiguousarray(lose[same_voxel])
    assert len(lose) > 500
    want = po.map_add(cur, lose, True, DS)
    same_points(want, cur, "the oracle drops them all")
    before = h.map_stats()
    h.map_add(lose, True, DS)
    same_points(h.map_download(), cur, "nothing survives")
    after = h.map_stats()
    assert after["brickwise"] == before["brickwise"] + 1 and after["reindex"] == before["reindex"]
    assert all(after[k] == before[k] for k in ("slots_used", "ids", "bricks"))
    one = np.ascontiguousarray(np.vstack([lose[:500], (cur.max(0) + np.float32(3.0))[None, :]]))
    cur2 = po.map_add(cur, one, True, DS)
    assert len(cur2) == len(cur) + 1
    h.map_add(one, True, DS)
    same_points(h.map_download(), cur2, "one survivor")
    search_matches(h, cur2, pr.body, pr.x_true)


def test_map_add_spanning_kilometres():
    """A change whose points are spread over kilometres: thousands of voxels and bricks with one point each (a voxel table
    with long probe sequences, brick keys that differ in every bit field), against the oracle, with and without the
    one-workgroup path."""
    rng = np.random.default_rng(9)
    box = np.array([600.0, 600.0, 300.0], np.float32)
    m = rng.uniform(-box, box, (30000, 3)).astype(np.float32)
    add = np.vstack([rng.uniform(-box, box, (5000, 3)), m[rng.integers(0, len(m), 2000)] + rng.normal(0, 0.1, (2000, 3))]).astype(np.float32)
    want = po.map_add(m, add, True, DS)
    for f in (1, 0):
        h = capi.Handle(fused_small_changes=f)
        h.map_build(m)
        h.map_add(add, True, DS)
        same_points(h.map_download(), want, f"fused={f}")
