"""Oracle checks for h_share_model and the IEKF loop (SURVEY.md 8c iii-v).  CPU only."""
import numpy as np
import pytest

from fast_lio_amd import synth
from oracle import pyoracle as po


@pytest.fixture(scope="module")
def small():
    pr = synth.make_problem(60000, 4000, "avia", cfg=101)
    m = po.Map(pr.map_xyz)
    xp, P = synth.propagate_prior_cov(po.predict, pr.x_prior)
    return pr, m, xp, P


def residuals_for(sc, m, x, sel_idx, planes):
    """Point-to-plane residual of the selected points for FIXED planes, as a function of the state."""
    xs = x
    R = synth.quat_to_R(xs[3:7])
    RLI = synth.quat_to_R(xs[7:11])
    pb = sc_body[sel_idx].astype(np.float64)
    pw = (R @ (RLI @ pb.T + xs[11:14, None]) + xs[0:3, None]).T
    return np.einsum("ij,ij->i", planes[:, :3].astype(np.float64), pw) + planes[:, 3]


def test_jacobian_rows_vs_finite_differences(small):
    global sc_body
    pr, m, xp, P = small
    sc = po.Scan(pr.body, nthreads=2)
    sc_body = pr.body
    for ext in (True, False):
        sc.reset()
        assert sc.h_share_model(m, xp, True, ext)
        sel = np.nonzero(sc.selected)[0]
        H = sc.h_x
        nv = sc.normvec[sel]
        assert H.shape == (len(sel), 12)
        # plane offset d = pd2 - n.p_w (so that the residual at xp equals pd2)
        pw = sc.world[sel].astype(np.float64)
        d = nv[:, 3].astype(np.float64) - np.einsum("ij,ij->i", nv[:, :3].astype(np.float64), pw)
        planes = np.concatenate([nv[:, :3].astype(np.float64), d[:, None]], axis=1)
        np.testing.assert_allclose(sc.h, -nv[:, 3], rtol=0, atol=0)
        eps = 1e-6
        ncols = 12 if ext else 6
        for c in range(ncols):
            dx = np.zeros(23)
            dx[c] = eps
            rp = residuals_for(sc, m, po.state_boxplus(xp, dx), sel, planes)
            rm = residuals_for(sc, m, po.state_boxplus(xp, -dx), sel, planes)
            fd = (rp - rm) / (2 * eps)
            # right-perturbation: d r/d(rot) = n^T R (-hat(p_I)) = (hat(p_I) R^T n)^T = A; likewise offR -> B
            np.testing.assert_allclose(H[:, c], fd, rtol=2e-5, atol=2e-5 * max(1.0, np.abs(fd).max()))
        if not ext:
            assert np.all(H[:, 6:] == 0.0)  # laserMapping.cpp:745: six literal zeros


def test_selection_semantics_between_passes(small):
    pr, m, xp, P = small
    sc = po.Scan(pr.body, nthreads=2)
    assert sc.h_share_model(m, xp, True, False)
    sel1 = sc.selected.copy()
    nn1 = sc.nn_idx.copy()
    n1 = sc.n_eff
    assert n1 == sel1.sum() > 100
    # no-search pass at a slightly different state: neighbours are reused, selection can only shrink
    x2 = po.state_boxplus(xp, np.r_[0.01, -0.01, 0.005, np.zeros(20)])
    assert sc.h_share_model(m, x2, False, False)
    np.testing.assert_array_equal(sc.nn_idx, nn1)
    sel2 = sc.selected
    assert np.all(sel2 <= sel1)
    # a search pass re-opens every point: identical to a fresh scan evaluated at x2
    assert sc.h_share_model(m, x2, True, False)
    fresh = po.Scan(pr.body, nthreads=2)
    fresh.h_share_model(m, x2, True, False)
    np.testing.assert_array_equal(sc.selected, fresh.selected)


def test_gate_sqdist_le_5_and_radius_bounded_equivalence(small):
    pr, m, xp, P = small
    sc = po.Scan(pr.body, nthreads=2)
    # push the scan 1.5 m up so many 5th neighbours straddle the d2 <= 5 gate
    xs = xp.copy()
    xs[2] += 1.9
    sc.h_share_model(m, xs, True, False)
    selA, HA, hA = sc.selected.copy(), sc.h_x.copy(), sc.h.copy()
    d5 = sc.nn_d2[:, 4]
    assert np.all(selA[d5 > 5] == 0)
    assert (d5 > 5).any() and (d5 <= 5).any()
    sc2 = po.Scan(pr.body, nthreads=2)
    sc2.set_search_radius2(5.0)
    sc2.h_share_model(m, xs, True, False)
    np.testing.assert_array_equal(sc2.selected, selA)
    np.testing.assert_array_equal(sc2.h_x, HA)
    np.testing.assert_array_equal(sc2.h, hA)


def test_no_effective_points_is_invalid(small):
    pr, m, xp, P = small
    sc = po.Scan(pr.body, nthreads=2)
    far = xp.copy()
    far[0:3] += 5000.0
    assert not sc.h_share_model(m, far, True, False)
    assert sc.n_eff == 0
    x, Pn, st = sc.update_iterated(m, far, P)
    assert st.passes == 4 and st.returned_in_loop == 0
    np.testing.assert_array_equal(x, far)  # every pass skipped: state untouched (esekfom.hpp:1638-1641)
    np.testing.assert_array_equal(Pn, P)


def test_info_form_equals_gain_form(small):
    pr, m, xp, P = small
    sc = po.Scan(pr.body, nthreads=2)
    sc.h_share_model(m, xp, True, True)
    H, h = sc.h_x, sc.h
    for n in (30, 200):  # both branches of esekfom.hpp:1715/1745 on the same rows
        Hs, hs = H[:n], h[:n]
        xa, Pa, Ka, da = po.iekf_pass_info(xp, xp, P, 0.001, Hs.T @ Hs, Hs.T @ hs)
        xb, Pb, Kb, db = po.iekf_pass_gain(xp, xp, P, 0.001, Hs, hs)
        np.testing.assert_allclose(da, db, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(Ka, Kb, rtol=1e-5, atol=1e-8)


def test_update_schedule_and_convergence(small):
    pr, m, xp, P = small
    sc = po.Scan(pr.body, nthreads=2)
    x, Pn, st = sc.update_iterated(m, xp, P)
    assert 2 <= st.passes <= 4 and st.returned_in_loop == 1
    assert st.pass_search[0] == 1
    e0 = po.state_boxminus(xp, pr.x_true)
    e1 = po.state_boxminus(x, pr.x_true)
    assert np.linalg.norm(e1[3:6]) < 0.2 * np.linalg.norm(e0[3:6])   # attitude pulled in
    assert abs(e1[2]) < 0.005                                        # height from the ground plane
    assert np.all(np.diag(Pn)[:6] < np.diag(P)[:6])
    assert np.allclose(Pn, Pn.T, atol=1e-6 * np.abs(Pn).max())


def test_max_iter_one_and_zero(small):
    pr, m, xp, P = small
    sc = po.Scan(pr.body, nthreads=2)
    x, Pn, st = sc.update_iterated(m, xp, P, max_iter=1)
    assert st.passes == 2 and st.returned_in_loop == 1
    sc.reset()
    x0, P0, st0 = sc.update_iterated(m, xp, P, max_iter=0)
    assert st0.passes == 1 and st0.returned_in_loop == 1  # i=-1 == maximum_iter-1: one pass, final-covariance branch
    assert not np.array_equal(x0, xp)


def test_map_incremental_classify_runs(small):
    pr, m, xp, P = small
    sc = po.Scan(pr.body, nthreads=2)
    x, Pn, st = sc.update_iterated(m, xp, P)
    world, cls = sc.map_incremental_classify(m, x)
    assert set(np.unique(cls)) <= {0, 1, 2}
    assert (cls == 0).sum() > 0  # most scan points fall in already-occupied voxels
