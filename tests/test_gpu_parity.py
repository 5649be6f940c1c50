"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Bars (BASELINE.json north_star): point_selected_surf bit-exact; neighbour indices bit-exact wherever
they can influence the result (5th neighbour inside the d2 <= 5 gate); fp32 planes bit-exact; normal
equations to 1e-10 relative (fp64 sums in a different order); posterior state within 1e-4 relative
and pose within 1e-4 m; covariance within 1e-4 of max|P| (norm-wise: P = L - K_x P cancels, see
tests/test_host_iekf.py).
"""
import numpy as np
import pytest

from fast_lio_amd import capi, synth
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

REL_NE = 1e-10   # normal equations, relative to max|entry|
REL_X = 1e-4     # state, relative
POSE_M = 1e-4    # pose, metres
REL_P = 1e-4     # covariance, relative to max|P|


@pytest.fixture(scope="module")
def prob():
    pr = synth.make_problem(200000, 20000, "avia", cfg=1)
    m = po.Map(pr.map_xyz)
    xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    return pr, m, xp, P, h


def check_eval(h, sc, m, x, search, ext, tag=""):
    HTH, HTh, n_eff, tres = h.eval(x, search, ext)
    valid = sc.h_share_model(m, x, search, ext)
    sel_g = h.fetch_selected()
    np.testing.assert_array_equal(sel_g, sc.selected, err_msg=f"{tag}: point_selected_surf differs")
    assert n_eff == sc.n_eff
    if valid:
        H0, h0 = sc.normal_equations()
        sH = np.abs(H0).max()
        np.testing.assert_allclose(HTH, H0, rtol=0, atol=REL_NE * sH, err_msg=tag)
        np.testing.assert_allclose(HTh, h0, rtol=0, atol=REL_NE * max(np.abs(h0).max(), 1e-30) + 1e-12 * np.sqrt(sH), err_msg=tag)
        assert abs(tres - sc.total_residual) <= 1e-9 * max(1.0, abs(sc.total_residual))
        np.testing.assert_allclose(HTH, HTH.T, rtol=0, atol=1e-12 * sH)
        if not ext:
            assert np.all(HTH[6:, :] == 0) and np.all(HTH[:, 6:] == 0) and np.all(HTh[6:] == 0)
    sel = sel_g.astype(bool)
    # fp32 artefacts are bit-exact
    np.testing.assert_array_equal(h.fetch_world().view(np.uint32), sc.world.view(np.uint32), err_msg=f"{tag}: world")
    np.testing.assert_array_equal(h.fetch_normvec()[sel].view(np.uint32), sc.normvec[sel].view(np.uint32), err_msg=f"{tag}: normvec")
    return n_eff


def check_neighbors(h, sc):
    idx, d2, cnt = h.fetch_neighbors()
    o_idx, o_d2, o_cnt = sc.nn_idx, sc.nn_d2, sc.nn_cnt
    gate = (o_cnt == 5) & (o_d2[:, 4] <= 5.0)  # every query whose result can matter downstream
    np.testing.assert_array_equal(idx[gate], o_idx[gate])
    np.testing.assert_array_equal(d2[gate].view(np.uint32), o_d2[gate].view(np.uint32))
    np.testing.assert_array_equal(cnt[gate], 5)
    # rejected queries: whatever the GPU kept must still be true neighbours in ascending order
    rej = ~gate
    assert np.all((cnt[rej] < 5) | (d2[rej][:, 4] > 5.0))
    return int(gate.sum())


@pytest.mark.parametrize("ext", [False, True])
def test_eval_search_and_nosearch_passes(prob, ext):
    pr, m, xp, P, h = prob
    h.scan_upload(pr.body)
    sc = po.Scan(pr.body, nthreads=8)
    n1 = check_eval(h, sc, m, xp, True, ext, "search@prior")
    assert n1 > 1000
    assert check_neighbors(h, sc) > 1000
    x2 = po.state_boxplus(xp, np.r_[0.02, -0.01, 0.015, 0.002, -0.001, 0.003, np.zeros(17)])
    check_eval(h, sc, m, x2, False, ext, "no-search@x2")
    check_eval(h, sc, m, pr.x_true, False, ext, "no-search@truth")
    check_eval(h, sc, m, pr.x_true, True, ext, "search@truth")
    check_neighbors(h, sc)


@pytest.mark.parametrize("lpq,sort,one_launch", [(0, 1, 0), (4, 0, 0), (4, 1, 0), (4, 0, 1), (4, 1, 1)])
def test_search_kernel_variants(prob, lpq, sort, one_launch):
    """The general exact kernel for every query, and the ring search as three launches / as ONE launch (flh_config.pass_kernel),
    with and without the Morton order of the scan: the same neighbours, flags, planes and normal equations as the oracle."""
    pr, m, xp, P, _ = prob
    h = capi.Handle(lanes_per_query=lpq, sort_queries=sort, pass_kernel=one_launch)
    h.map_build(pr.map_xyz)
    h.scan_upload(pr.body[:5000])
    sc = po.Scan(pr.body[:5000], nthreads=8)
    check_eval(h, sc, m, xp, True, False, f"lpq={lpq}")
    check_neighbors(h, sc)
    h.close()


def test_fetch_rows_matches_oracle(prob):
    pr, m, xp, P, h = prob
    h.scan_upload(pr.body)
    sc = po.Scan(pr.body, nthreads=8)
    for ext in (False, True):
        h.eval(xp, True, ext)
        sc.h_share_model(m, xp, True, ext)
        Hx, hv = h.fetch_rows()
        assert Hx.shape == sc.h_x.shape
        np.testing.assert_allclose(Hx, sc.h_x, rtol=1e-13, atol=1e-13)
        np.testing.assert_array_equal(hv, sc.h)


@pytest.mark.parametrize("ext", [False, True])
def test_full_update_matches_oracle(prob, ext):
    pr, m, xp, P, h = prob
    h.scan_upload(pr.body)
    kf = capi.Esekf(h, max_iter=3, extrinsic_est_en=ext)
    kf.change_x(xp)
    kf.change_P(P)
    st = kf.update(0.001)
    sc = po.Scan(pr.body, nthreads=8)
    x_ref, P_ref, st_ref = sc.update_iterated(m, xp, P, extrinsic_est_en=ext)
    assert st.passes == st_ref.passes and st.searches == st_ref.searches
    assert list(st.pass_search)[: st.passes] == list(st_ref.pass_search)[: st.passes]
    assert list(st.n_eff)[: st.passes] == list(st_ref.n_eff)[: st.passes]
    x, Pn = kf.get_x(), kf.get_P()
    assert np.linalg.norm(x[0:3] - x_ref[0:3]) <= POSE_M
    np.testing.assert_allclose(x, x_ref, rtol=REL_X, atol=1e-7)
    np.testing.assert_allclose(Pn, P_ref, rtol=0, atol=REL_P * np.abs(P_ref).max())
    np.testing.assert_array_equal(h.fetch_selected(), sc.selected)
    # run-to-run determinism of the GPU path (fixed reduction order)
    h.scan_upload(pr.body)
    kf.change_x(xp)
    kf.change_P(P)
    kf.update(0.001)
    np.testing.assert_array_equal(kf.get_x(), x)
    np.testing.assert_array_equal(kf.get_P(), Pn)


def test_prelaunched_nosearch_pass_same_bits(prob):
    """flh_eval_expect_next (the mirror filter announces a no-search pass; its kernel is enqueued beside the pass before it and takes
    its state from a mailbox): (0) kernel against kernel at one state, (1) the iterated update with the hints honoured against the
    oracle and, bit for bit, against the same update with flh_config.prelaunch = 0, (2) a host that comes after the waiting kernel
    has given up (20 ms): the pass is launched the usual way, same bits; the counters show that the mailbox was really used."""
    import time

    pr, m, xp, P, _ = prob
    ha, hb = capi.Handle(prelaunch=0), capi.Handle(prelaunch=1)
    for hh in (ha, hb):
        hh.map_build(pr.map_xyz)
        hh.scan_upload(pr.body)
        hh.set_timing_stride(0)  # (a timed evaluation is never handed to the mailbox)
    # (0) one evaluation
    ref_s, ref_n = ha.eval(xp, True, False), ha.eval(pr.x_true, False, False)
    hb.expect_next(1)
    got_s = hb.eval(xp, True, False)        # enqueues the next pass's kernel behind its own
    c0 = hb.prelaunch_stats()
    got_n = hb.eval(pr.x_true, False, False)  # handed over through the mailbox
    c1 = hb.prelaunch_stats()
    assert c0["armed"] == 1 and c1["go"] == 1 and c1["abort"] == 0
    for a, b in ((got_s, ref_s), (got_n, ref_n)):
        np.testing.assert_array_equal(a[0], b[0]); np.testing.assert_array_equal(a[1], b[1])
        assert a[2] == b[2] and a[3] == b[3]
    np.testing.assert_array_equal(hb.fetch_selected(), ha.fetch_selected())
    # a hint that does not come true: the waiting kernel is released, the search runs the usual way
    hb.expect_next(1)
    hb.eval(xp, True, False)
    got_s2 = hb.eval(xp, True, False)
    np.testing.assert_array_equal(got_s2[0], ref_s[0])
    assert hb.prelaunch_stats()["abort"] == 1
    assert ha.prelaunch_stats() == {"armed": 0, "go": 0, "abort": 0, "gone": 0}
    ha.expect_next(1); ha.eval(xp, True, False)
    assert ha.prelaunch_stats()["armed"] == 0  # prelaunch = 0: hints are ignored
    # (1) the whole update, both extrinsic settings, against the oracle and against the plain launches
    sc = po.Scan(pr.body, nthreads=8)
    for ext in (False, True):
        res = []
        for hh in (ha, hb):
            hh.scan_upload(pr.body)
            kf = capi.Esekf(hh, max_iter=3, extrinsic_est_en=ext)
            kf.change_x(xp); kf.change_P(P)
            st = kf.update(0.001)
            res.append((kf.get_x().copy(), kf.get_P().copy(), st.passes, list(st.n_eff)[: st.passes], list(st.pass_search)[: st.passes],
                        hh.fetch_selected().copy()))
            kf.close()
        a, b = res
        assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes() and a[2:5] == b[2:5]
        np.testing.assert_array_equal(a[5], b[5])
        x_ref, P_ref, st_ref = sc.update_iterated(m, xp, P, extrinsic_est_en=ext)
        assert b[2] == st_ref.passes and b[3] == list(st_ref.n_eff)[: st_ref.passes]
        assert np.linalg.norm(b[0][0:3] - x_ref[0:3]) <= POSE_M
        np.testing.assert_allclose(b[0], x_ref, rtol=REL_X, atol=1e-7)
        np.testing.assert_allclose(b[1], P_ref, rtol=0, atol=REL_P * np.abs(P_ref).max())
        np.testing.assert_array_equal(b[5], sc.selected)
    c2 = hb.prelaunch_stats()
    assert c2["go"] > c1["go"], c2  # the filter's hints put the no-search passes through the mailbox
    # (2) lateness
    hb.scan_upload(pr.body)
    hb.expect_next(1)
    o_s = hb.eval(xp, True, False)
    time.sleep(0.05)                       # the waiting kernel gives up after 20 ms
    o_n = hb.eval(pr.x_true, False, False)  # mail posted to nobody: flh_eval_end notices and launches the usual way
    c3 = hb.prelaunch_stats()
    assert c3["gone"] == c2["gone"] + 1
    np.testing.assert_array_equal(o_s[0], ref_s[0]); np.testing.assert_array_equal(o_n[0], ref_n[0]); np.testing.assert_array_equal(o_n[1], ref_n[1])
    assert o_n[2] == ref_n[2]
    # nothing follows: a waiting kernel is released at once and other calls go on as usual
    hb.expect_next(1); hb.eval(xp, True, False); hb.expect_next(2)
    assert hb.prelaunch_stats()["abort"] == c3["abort"] + 1
    ha.scan_upload(pr.body); ha.eval(xp, True, False)
    np.testing.assert_array_equal(hb.fetch_selected(), ha.fetch_selected())
    ha.close(); hb.close()


@pytest.mark.parametrize("ext", [False, True])
def test_reference_operation_sequence_of_the_filter_on_gpu_normal_equations(prob, ext):
    """The product's filter forms the information-form step as one 12 x 12 elimination (include/fastlio_amd/esekfom.hpp, info_cols)
    -- a deviation from the reference's two 23 x 23 inverses (esekfom.hpp:1782-1802), named in INTEGRATION.md.  Here the SAME filter
    built with -DFASTLIO_AMD_REFERENCE_ALGEBRA (the reference's sequence; _build.build_reference_algebra) runs on the GPU's normal
    equations: its posterior must sit on the oracle's -- which follows the reference's sequence too -- at the tolerances the
    reference's sequence resolves, and the product's form must sit where that build sits."""
    from fast_lio_amd import _build

    pr, m, xp, P, h = prob
    h.scan_upload(pr.body)
    sc = po.Scan(pr.body, nthreads=8)
    x_ref, P_ref, st_ref = sc.update_iterated(m, xp, P, extrinsic_est_en=ext)

    def model(x, converge):
        HTH, HTh, n_eff, tres = h.eval(x, converge, ext)
        return {"valid": n_eff > 0, "n_eff": n_eff, "HTH": HTH, "HTh": HTh, "total_residual": tres}

    with capi.using_library(_build.build_reference_algebra()):
        kf = capi.Esekf(None, max_iter=3, extrinsic_est_en=ext)
        kf.set_meas_model(model)
        kf.change_x(xp); kf.change_P(P)
        st = kf.update(0.001)
        x_r, P_r = kf.get_x().copy(), kf.get_P().copy()
        kf.close()
    assert st.passes == st_ref.passes and list(st.n_eff)[: st.passes] == list(st_ref.n_eff)[: st_ref.passes]
    h.scan_upload(pr.body)
    kf = capi.Esekf(h, max_iter=3, extrinsic_est_en=ext)
    kf.change_x(xp); kf.change_P(P)
    kf.update(0.001)
    x_p, P_p = kf.get_x().copy(), kf.get_P().copy()
    kf.close()
    pmax = np.abs(P_ref).max()
    d_ref = (np.abs(x_r - x_ref).max(), np.abs(P_r - P_ref).max() / pmax)
    d_prod = (np.abs(x_p - x_ref).max(), np.abs(P_p - P_ref).max() / pmax)
    print(f"[reference-sequence filter on GPU normal equations, ext={ext}] vs oracle: |dx| {d_ref[0]:.2e}, |dP|/max|P| {d_ref[1]:.2e}; "
          f"product's 12x12 form vs oracle: |dx| {d_prod[0]:.2e}, |dP|/max|P| {d_prod[1]:.2e}")
    # without extrinsic estimation the reference's sequence resolves 1e-11 m on the state (measured on the MI355X: 4.1e-13) and
    # 1e-9 max|P| on the covariance (measured: 3.8e-11; P = L - K_x P cancels); with it the double inversion amplifies the last bits
    # of the normal equations (tests/test_host_algebra.py: 1e-9 .. 1e-8 m)
    tol_x, tol_P = (1e-7, 1e-5) if ext else (1e-11, 1e-9)
    assert d_ref[0] <= tol_x and d_ref[1] <= tol_P, d_ref
    assert d_prod[0] <= max(10 * tol_x, 1e-10) and d_prod[1] <= max(10 * tol_P, 1e-8), d_prod


def test_staged_scan_ring_equals_direct_upload(prob):
    pr, m, xp, P, h = prob
    h.scan_upload(pr.body)
    a = h.eval(xp, True, False)
    h.scan_stage(3, pr.body)
    h.scan_stage(4, pr.body[:777])
    h.scan_activate(4)
    assert h.N == 777
    h.eval(xp, True, False)
    h.scan_activate(3)
    b = h.eval(xp, True, False)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    assert a[2] == b[2]


# ------------------------------------------------------------------ kNN on adversarial maps
def test_knn_lattice_ties_and_duplicates():
    rng = np.random.default_rng(7)
    g = np.arange(-6, 7, dtype=np.float32) * 0.5
    lat = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    lat = np.concatenate([lat, lat[:300]], axis=0)  # exact duplicates: ties resolved by lower map index
    lat = lat[rng.permutation(len(lat))]
    m = po.Map(lat)
    h = capi.Handle()
    h.map_build(lat)
    q = (rng.integers(-6, 7, (3000, 3)) * 0.5 + rng.choice([0.0, 0.25], (3000, 3))).astype(np.float32)
    h.scan_upload(q)
    ident = synth.make_state()
    h.eval(ident, True, False)
    sc = po.Scan(q, nthreads=8)
    sc.h_share_model(m, ident, True, False)
    check_neighbors(h, sc)
    np.testing.assert_array_equal(h.fetch_selected(), sc.selected)
    h.close()


def test_knn_sparse_map_needs_ring_expansion():
    # spacing 1.3 m: the 5th neighbour is usually outside the 3x3x3 stencil's guaranteed radius
    rng = np.random.default_rng(11)
    pts = (rng.uniform(-40, 40, (12000, 3)) * np.array([1, 1, 0.05])).astype(np.float32)
    m = po.Map(pts)
    h = capi.Handle()
    h.map_build(pts)
    q = (rng.uniform(-45, 45, (8000, 3)) * np.array([1, 1, 0.08])).astype(np.float32)
    h.scan_upload(q)
    ident = synth.make_state()
    h.eval(ident, True, False)
    sc = po.Scan(q, nthreads=8)
    sc.h_share_model(m, ident, True, False)
    n = check_neighbors(h, sc)
    assert n > 100
    np.testing.assert_array_equal(h.fetch_selected(), sc.selected)
    h.close()


def test_knn_dense_cells_and_large_coordinates():
    # 40 points per cell on average, coordinates near +-1900 m (fp32 ulp ~1e-4)
    rng = np.random.default_rng(13)
    pts = (rng.uniform(0, 12, (70000, 3)) + np.array([1900.0, -1900.0, 30.0])).astype(np.float32)
    m = po.Map(pts)
    h = capi.Handle()
    h.map_build(pts)
    q = (rng.uniform(-1, 13, (4000, 3)) + np.array([1900.0, -1900.0, 30.0])).astype(np.float32)
    h.scan_upload(q)
    ident = synth.make_state()
    h.eval(ident, True, False)
    sc = po.Scan(q, nthreads=8)
    sc.h_share_model(m, ident, True, False)
    check_neighbors(h, sc)
    np.testing.assert_array_equal(h.fetch_selected(), sc.selected)
    h.close()


# ------------------------------------------------------------------ edge cases
def test_edge_empty_scan_and_tiny_maps(prob):
    pr, m, xp, P, _ = prob
    h = capi.Handle()
    h.map_build(pr.map_xyz[:3])          # fewer than 5 map points: nothing can be selected
    h.scan_upload(pr.body[:100])
    HTH, HTh, n_eff, tres = h.eval(xp, True, False)
    assert n_eff == 0 and np.all(HTH == 0) and np.all(HTh == 0) and tres == 0
    assert h.fetch_selected().sum() == 0
    idx, d2, cnt = h.fetch_neighbors()
    assert cnt.max() <= 3
    h.map_build(np.zeros((0, 3), np.float32))  # empty map
    h.scan_upload(pr.body[:100])
    assert h.eval(xp, True, False)[2] == 0
    h.map_build(pr.map_xyz)
    h.scan_upload(np.zeros((0, 3), np.float32))  # empty scan
    assert h.eval(xp, True, False)[2] == 0
    for n in (1, 63, 64, 65, 257):         # ragged sizes around the wave / block width
        h.scan_upload(pr.body[:n])
        sc = po.Scan(pr.body[:n], nthreads=1)
        check_eval(h, sc, m, xp, True, False, f"N={n}")
    h.close()


def test_edge_no_effective_points_and_gain_form(prob):
    pr, m, xp, P, h = prob
    far = xp.copy()
    far[0:3] += 3000.0
    h.scan_upload(pr.body)
    kf = capi.Esekf(h, max_iter=3)
    kf.change_x(far)
    kf.change_P(P)
    st = kf.update(0.001)
    assert st.passes == 4 and st.returned_in_loop == 0 and list(st.n_eff)[:4] == [0, 0, 0, 0]
    np.testing.assert_array_equal(kf.get_x(), far)     # every pass skipped (esekfom.hpp:1638-1641)
    np.testing.assert_array_equal(kf.get_P(), P)
    # fewer than 23 effective points -> gain-form branch through flh_fetch_rows
    body = pr.body[:16]
    h.scan_upload(body)
    kf.change_x(xp)
    kf.change_P(P)
    st = kf.update(0.001)
    sc = po.Scan(body, nthreads=1)
    x_ref, P_ref, st_ref = sc.update_iterated(m, xp, P)
    assert 0 < max(st_ref.n_eff) < 23
    assert list(st.n_eff)[: st.passes] == list(st_ref.n_eff)[: st_ref.passes]
    np.testing.assert_allclose(kf.get_x(), x_ref, rtol=REL_X, atol=1e-7)
    np.testing.assert_allclose(kf.get_P(), P_ref, rtol=0, atol=REL_P * np.abs(P_ref).max())


def test_nosearch_before_search_is_an_error(prob):
    pr, m, xp, P, h = prob
    h.scan_upload(pr.body[:100])
    with pytest.raises(capi.FlhError, match="do_search == 0 before any search"):
        h.eval(xp, False, False)


def test_map_extent_limit_is_reported():
    h = capi.Handle()
    pts = np.array([[0, 0, 0], [9000, 0, 0], [1, 1, 1], [2, 2, 2], [3, 3, 3]], np.float32)
    with pytest.raises(capi.FlhError, match="4096 cells"):
        h.map_build(pts)
    h2 = capi.Handle(cell_size=4.0)
    h2.map_build(pts)
    h.close()
    h2.close()


@pytest.mark.parametrize("sensor,M,N", [("velodyne", 150000, 12000), ("ouster64", 150000, 15000), ("mid360", 120000, 9000)])
def test_other_sensors_full_update(sensor, M, N):
    pr = synth.make_problem(M, N, sensor, cfg=3)
    m = po.Map(pr.map_xyz)
    xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    h.scan_upload(pr.body)
    kf = capi.Esekf(h, max_iter=3)
    kf.change_x(xp)
    kf.change_P(P)
    st = kf.update(0.001)
    sc = po.Scan(pr.body, nthreads=8)
    x_ref, P_ref, st_ref = sc.update_iterated(m, xp, P)
    assert list(st.n_eff)[: st.passes] == list(st_ref.n_eff)[: st_ref.passes]
    assert np.linalg.norm(kf.get_x()[0:3] - x_ref[0:3]) <= POSE_M
    np.testing.assert_allclose(kf.get_x(), x_ref, rtol=REL_X, atol=1e-7)
    np.testing.assert_allclose(kf.get_P(), P_ref, rtol=0, atol=REL_P * np.abs(P_ref).max())
    np.testing.assert_array_equal(h.fetch_selected(), sc.selected)
    h.close()


def test_cell_size_variants(prob):
    pr, m, xp, P, _ = prob
    for c in (0.6, 1.0, 1.5, 3.0):
        h = capi.Handle(cell_size=c)
        h.map_build(pr.map_xyz)
        h.scan_upload(pr.body[:6000])
        sc = po.Scan(pr.body[:6000], nthreads=8)
        check_eval(h, sc, m, xp, True, False, f"cell={c}")
        check_neighbors(h, sc)
        h.close()


def test_eval_without_scan_or_map_fails_loudly(prob):
    """flh_eval before a scan / a map must return an error, not fault on the device."""
    pr, m, xp, P, _ = prob
    h = capi.Handle()
    with pytest.raises(capi.FlhError, match="no active scan"):
        h.eval(xp, True, False)
    h.scan_upload(pr.body[:100])
    with pytest.raises(capi.FlhError, match="no map"):
        h.eval(xp, True, False)
    h.map_build(pr.map_xyz)
    h.eval(xp, True, False)
    kf = capi.Esekf(None)
    kf.change_x(xp)
    kf.change_P(P)
    with pytest.raises(capi.FlhError, match="no flh_handle|no device handle"):
        kf.update(0.001)
    h.close()


def test_async_staging_equals_synchronous_staging(prob):
    pr, m, xp, P, _ = prob
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    bodies = [np.ascontiguousarray(pr.body[i::3][:4000]) for i in range(3)]
    ref = []
    for b in bodies:
        h.scan_upload(b)
        ref.append((h.eval(xp, True, False), h.fetch_selected(), h.fetch_neighbors()[0]))
    for s, b in enumerate(bodies):          # all three in flight on the staging thread at once
        h.scan_stage_async(s, b)
    for s in (2, 0, 1):
        h.scan_activate(s)
        got = h.eval(xp, True, False)
        np.testing.assert_array_equal(got[0], ref[s][0][0])
        np.testing.assert_array_equal(got[1], ref[s][0][1])
        assert got[2] == ref[s][0][2]
        np.testing.assert_array_equal(h.fetch_selected(), ref[s][1])
        np.testing.assert_array_equal(h.fetch_neighbors()[0], ref[s][2])
        np.testing.assert_array_equal(h.fetch_scan().view(np.uint32), bodies[s].view(np.uint32))
    h.scan_stage_async(5, bodies[0])
    h.scan_wait(5)
    with pytest.raises(capi.FlhError):
        h.scan_stage(3, np.zeros((4, 3), np.float32)[:, :2])  # stride < 12
    h.close()


def test_frame_world_and_points_body_to_world(prob):
    """publish_frame_world's RGBpointBodyToWorld loops (src/laserMapping.cpp:478-530, :200-211)."""
    pr, m, xp, P, _ = prob
    h = capi.Handle()
    x = pr.x_true.copy()
    x[7:11] = (0.01, -0.02, 0.03, 0.9993)
    x[7:11] /= np.linalg.norm(x[7:11])
    rng = np.random.default_rng(3)
    cloud = (np.repeat(pr.body[:15000], 2, axis=0) + rng.normal(0, 0.05, (30000, 3))).astype(np.float32)
    want = po.points_body_to_world(x, cloud)
    np.testing.assert_array_equal(h.points_body_to_world(x, cloud).view(np.uint32), want.view(np.uint32))
    wide = np.zeros((len(cloud), 12), np.float32)   # pcl::PointXYZINormal records
    wide[:, :3] = cloud
    wide[:, 4:] = 7.0
    np.testing.assert_array_equal(h.points_body_to_world(x, wide).view(np.uint32), want.view(np.uint32))
    # the device-resident clouds of a staged raw scan: feats_undistort (dense) and feats_down_body
    n = h.scan_stage_downsampled(2, cloud, 0.5)
    np.testing.assert_array_equal(h.frame_world(x, slot=2, dense=True).view(np.uint32), want.view(np.uint32))
    down = po.voxel_grid(cloud, 0.5)
    assert n == len(down)
    np.testing.assert_array_equal(h.frame_world(x, slot=2, dense=False).view(np.uint32),
                                  po.points_body_to_world(x, down).view(np.uint32))
    h.scan_stage(3, cloud[:100])
    with pytest.raises(capi.FlhError, match="not staged from a raw scan"):
        h.frame_world(x, slot=3, dense=True)
    assert h.points_body_to_world(x, cloud[:0]).shape == (0, 3)
    h.close()


def test_rccl_allreduce_path_single_rank(prob):
    """The native multi-GPU path (flh_rccl_*): with a communicator attached, flh_eval leaves its Gram block on the device,
    RCCL sums it over the ranks and a publish kernel hands it to the host.  The searching pass is the ONE-launch pass there too:
    the group reducers leave their totals in device memory and the last group adds the groups in the order the host adds
    granules (groups_sum_device), so with one rank (the box has one GPU) the result must be the plain path's BIT FOR BIT,
    through eval, the full update and flh_eval_group."""
    pr, m, xp, P, _ = prob
    body = pr.body[:6000]
    ref_h = capi.Handle()
    ref_h.map_build(pr.map_xyz)
    ref_h.scan_upload(body)
    ref = ref_h.eval(xp, True, False)
    kf0 = capi.Esekf(ref_h, max_iter=3)
    kf0.change_x(xp); kf0.change_P(P)
    st0 = kf0.update(0.001)
    h = capi.Handle()
    h.rccl_init_rank(1, capi.rccl_unique_id(), 0)
    assert h.rccl_size() == 1
    h.map_build(pr.map_xyz)
    h.scan_upload(body)
    def close(a, b):
        np.testing.assert_allclose(a[0], b[0], rtol=0, atol=1e-12 * np.abs(b[0]).max())
        np.testing.assert_allclose(a[1], b[1], rtol=0, atol=1e-12 * np.abs(b[1]).max())
        assert a[2] == b[2] and abs(a[3] - b[3]) <= 1e-12 * max(abs(b[3]), 1.0)

    got = h.eval(xp, True, False)
    close(got, ref)
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)  # the granules' tree, added on the device
    got2 = h.eval(xp, False, False)
    np.testing.assert_array_equal(got2[0], got[0])  # search / no-search at one state: the same bits
    st_ = h.pass_stats()
    assert st_["search_passes"] == 1 and st_["one_launch_passes"] == 1
    kf = capi.Esekf(h, max_iter=3)
    kf.change_x(xp); kf.change_P(P)
    st = kf.update(0.001)
    assert st.passes == st0.passes and list(st.n_eff) == list(st0.n_eff)
    np.testing.assert_array_equal(kf.get_x(), kf0.get_x())
    np.testing.assert_array_equal(kf.get_P(), kf0.get_P())
    h.close()
    # one process, one handle per device
    g = capi.Handle()
    capi.rccl_init_all([g])
    g.map_build(pr.map_xyz)
    g.scan_upload(body)
    grp = capi.eval_group([g], xp, True, False)
    for a, b in zip(grp, got):
        np.testing.assert_array_equal(a, b)
    g.close()
    ref_h.close()


def test_peer_granules_one_and_two_handles(prob):
    """The exchange without a collective (flh_peer_*): every handle's group reducers write their granules into every handle's
    pinned buffer, the host adds (rank, group) in order.  One handle attached to itself: the plain path's bits.  One scan
    split over two handles (Morton-first shards, map replicated) through flh_eval_group: flags of every point as in the
    single-handle evaluation, normal equations equal up to the order of the fp64 sums, at a searching and a no-search pass,
    with and without the extrinsic columns."""
    from fast_lio_amd import dist as fdist

    pr, m, xp, P, _ = prob
    body = pr.body
    one = capi.Handle()
    one.map_build(pr.map_xyz)
    one.scan_upload(body)
    solo = capi.Handle()
    capi.peer_init_all([solo])
    assert solo.peer_size() == 1
    solo.map_build(pr.map_xyz)
    solo.scan_upload(body)
    shards = [fdist.morton_shard(body, r, 2) for r in range(2)]
    hs = [capi.Handle(), capi.Handle()]
    capi.peer_init_all(hs)
    assert hs[1].peer_size() == 2
    for hh, idx in zip(hs, shards):
        hh.map_build(pr.map_xyz)
        hh.scan_upload(np.ascontiguousarray(body[idx]))
    for ext in (False, True):
        for x, search in ((xp, True), (pr.x_true, False), (pr.x_true, True), (xp, False)):
            ref = one.eval(x, search, ext)
            got1 = capi.eval_group([solo], x, search, ext)
            for a, b in zip(got1, ref):
                np.testing.assert_array_equal(a, b)
            got = capi.eval_group(hs, x, search, ext)
            np.testing.assert_allclose(got[0], ref[0], rtol=0, atol=1e-12 * np.abs(ref[0]).max())
            np.testing.assert_allclose(got[1], ref[1], rtol=0, atol=1e-12 * np.abs(ref[1]).max())
            assert got[2] == ref[2] and abs(got[3] - ref[3]) <= 1e-12 * max(abs(ref[3]), 1.0)
            sel = one.fetch_selected()
            for hh, idx in zip(hs, shards):
                np.testing.assert_array_equal(hh.fetch_selected(), sel[idx])
    for hh in hs + [solo, one]:
        hh.close()


def test_map_partitioned_over_two_handles_equals_the_whole_map(prob):
    """BASELINE configs[4] in miniature: the map cut into two slabs (+ halo), each on its own handle with the WHOLE scan and
    an owned interval; the sum of the two Gram blocks drives the same host filter.  Flags, n_eff per pass: identical to the
    single-handle update; posterior: equal up to the order of the fp64 sums."""
    from fast_lio_amd import dist as fdist

    pr, m, xp, P, _ = prob
    body = pr.body
    one = capi.Handle()
    one.map_build(pr.map_xyz)
    one.scan_upload(body)
    kf1 = capi.Esekf(one, max_iter=3)
    kf1.change_x(xp); kf1.change_P(P)
    st1 = kf1.update(0.001)
    sel1 = one.fetch_selected()
    axis, edges = fdist.partition_bounds(pr.map_xyz, 2)
    parts = []
    for r in range(2):
        keep = fdist.partition_slab(pr.map_xyz, axis, edges, r, fdist.HALO_DEFAULT)
        hh = capi.Handle()
        hh.map_build(pr.map_xyz[keep])
        hh.set_owned_interval(axis, edges[r], edges[r + 1])
        hh.scan_upload(body)
        parts.append(hh)
        assert hh.M < 0.75 * one.M

    def model(x, converge):
        tot = None
        for hh in parts:
            HTH, HTh, n_eff, tres = hh.eval(x, converge, False)
            cur = [HTH.copy(), HTh.copy(), n_eff, tres]
            tot = cur if tot is None else [tot[0] + cur[0], tot[1] + cur[1], tot[2] + cur[2], tot[3] + cur[3]]
        if tot[2] < 1:
            return {"valid": False, "n_eff": 0}
        return {"valid": True, "n_eff": tot[2], "HTH": tot[0], "HTh": tot[1], "total_residual": tot[3]}

    kf = capi.Esekf(None, max_iter=3)
    kf.set_meas_model(model)
    kf.change_x(xp); kf.change_P(P)
    st = kf.update(0.001)
    assert st.passes == st1.passes and list(st.n_eff)[: st.passes] == list(st1.n_eff)[: st1.passes]
    sels = [hh.fetch_selected() for hh in parts]
    assert not np.any(sels[0] & sels[1])                       # every point has one owner
    np.testing.assert_array_equal(sels[0] | sels[1], sel1)     # and the same verdict as on the whole map
    np.testing.assert_allclose(kf.get_x(), kf1.get_x(), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(kf.get_P(), kf1.get_P(), rtol=0, atol=1e-9 * np.abs(kf1.get_P()).max())
    for hh in parts:
        hh.close()
    one.close()


def test_fp16_plane_fit_ablation_is_close_but_not_the_default(prob):
    """BASELINE configs[4]'s ablation: plane_fit_dtype = 1 runs the 5x3 QR in fp16 on coordinates moved next to the plane.
    It is an approximation: ~90-95 % of the flags agree (measured: 5-9 % differ, tools/fp16_ablation.py), the planes both
    accept agree to a fraction of a degree, the posterior to millimetres -- and the default path stays the bit-exact fp32 one."""
    pr, m, xp, P, h32 = prob
    body = pr.body
    h32.scan_upload(body)
    ref = h32.eval(xp, True, False)
    sel32, nv32 = h32.fetch_selected().astype(bool), h32.fetch_normvec()
    h16 = capi.Handle(plane_fit_dtype=1)
    h16.map_build(pr.map_xyz)
    h16.scan_upload(body)
    got = h16.eval(xp, True, False)
    sel16, nv16 = h16.fetch_selected().astype(bool), h16.fetch_normvec()
    mismatch = float((sel16 != sel32).mean())
    assert 0.0 < mismatch < 0.15, mismatch                       # differs (it is fp16): a few percent of the flags
    both = sel16 & sel32
    cosang = np.abs(np.sum(nv16[both, :3] * nv32[both, :3], axis=1))
    assert np.median(np.degrees(np.arccos(np.clip(cosang, -1, 1)))) < 3.0   # half has 11 bits: ~0.5-1 degree on 0.7 m wide neighbourhoods
    assert abs(got[2] - ref[2]) < 0.15 * ref[2]
    kf16, kf32 = capi.Esekf(h16, max_iter=3), capi.Esekf(h32, max_iter=3)
    for kf in (kf16, kf32):
        kf.change_x(xp); kf.change_P(P)
        kf.update(0.001)
    assert np.linalg.norm(kf16.get_x()[:3] - kf32.get_x()[:3]) < 0.1      # centimetres: the price of 11-bit planes
    h16.close()


def test_run_scans_native_loop_equals_scan_by_scan_updates(prob):
    """flh_esekf_run_scans (the node's main loop run natively, next scan staged while this one updates) ends in the same
    posterior as update_scan called scan by scan, from host buffers and from pre-staged slots, in one call or chained."""
    pr, m, xp, P, _ = prob
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    bodies = [np.ascontiguousarray(pr.body[i::4][:3000]) for i in range(4)]
    xs = np.ascontiguousarray(xp, np.float64)
    Ps = np.ascontiguousarray(P, np.float64)
    priors = [(xs, Ps)] * 4
    kf = capi.Esekf(h, max_iter=3)
    want = []
    for b in bodies:
        h.scan_upload(b)
        kf.change_x(xs); kf.change_P(Ps)
        kf.update(0.001)
        want.append(kf.get_x())
    jobs = capi.Esekf.make_jobs(bodies, priors)
    h.stage_stats(reset=True)
    rs = kf.run_scans(jobs, 0, 4, ring=3)
    assert rs.scans == 4 and rs.passes >= 8 and rs.n_search_passes + rs.n_nosearch_passes == rs.passes
    np.testing.assert_array_equal(kf.get_x(), want[3])
    sd = h.stage_stats()  # the developer counters of the hand-over (flh_debug_stage_stats): every scan staged once, activated once
    assert sd["jobs"] == 4 and sd["activations"] == 4 and sd["enq_us"] > 0 and sd["act_wait_max_us"] >= 0
    kf.run_scans(jobs, 0, 2, ring=2, stage_next=True)
    np.testing.assert_array_equal(kf.get_x(), want[1])
    kf.run_scans(jobs, 2, 1, ring=2, first_staged=True)
    np.testing.assert_array_equal(kf.get_x(), want[2])
    for s, b in enumerate(bodies):
        h.scan_stage(10 + s, b)
    jobs2 = capi.Esekf.make_jobs(bodies, priors, slots=[10, 11, 12, 13])
    kf.run_scans(jobs2, 0, 6)     # cycles: the 6th scan is bodies[1]
    np.testing.assert_array_equal(kf.get_x(), want[1])
    h.close()
