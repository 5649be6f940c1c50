"""The CPU oracle against INDEPENDENT implementations (scipy / numpy), not against its own brute-force twin: a third party's
k-d tree for the 5-NN, scipy's Rotation for SO(3) and the body->world transform, numpy's least squares for the plane, a
dictionary-based voxel filter for Add_Points.  These do not pin the oracle to the reference's bits (nothing here can: Eigen,
PCL and ikd-Tree are absent) -- they pin it to the mathematics the cited reference lines implement, through code
that shares nothing with the restatement.  CPU only."""
import numpy as np
import pytest
from scipy.spatial import cKDTree
from scipy.spatial.transform import Rotation

from fast_lio_amd import synth
from oracle import pyoracle as po

RNG = np.random.default_rng(20240807)


def test_knn5_matches_scipy_ckdtree():
    """ikdtree.Nearest_Search (src/laserMapping.cpp:666): same five neighbours in the same order as scipy's k-d tree (fp64 on the
    same fp32 coordinates); squared distances equal to fp32 rounding.  Queries on and off the map, clustered and sparse."""
    pts = np.vstack([RNG.uniform(-30, 30, (20000, 3)), RNG.normal(0, 0.3, (5000, 3)) + [5, 5, 1]]).astype(np.float32)
    m = po.Map(pts)
    t = cKDTree(pts.astype(np.float64))
    q = np.vstack([RNG.uniform(-35, 35, (3000, 3)), RNG.normal(0, 0.5, (1000, 3)) + [5, 5, 1]]).astype(np.float32)
    idx, d2, cnt = m.knn5_batch(q, nthreads=4)
    dd, ii = t.query(q.astype(np.float64), k=5)
    assert np.all(cnt == 5)
    np.testing.assert_allclose(d2, dd ** 2, rtol=2e-6, atol=1e-9)
    # indices: identical wherever the five distances (and the sixth) are distinct beyond fp32 resolution
    d6 = t.query(q.astype(np.float64), k=6)[0] ** 2
    gaps = np.diff(d6, axis=1).min(axis=1) > 1e-5 * np.maximum(d6[:, 5], 1e-6)
    assert gaps.mean() > 0.9
    np.testing.assert_array_equal(idx[gaps], ii[gaps])


def test_so3_and_body_to_world_match_scipy_rotation():
    """SO3::exp / log (IKFoM mtk) and pointBodyToWorld (src/laserMapping.cpp:656-660) against scipy.spatial.transform."""
    for _ in range(200):
        v = RNG.normal(size=3) * RNG.choice([1e-6, 1e-2, 1.0, 3.0])
        q = po.so3_exp(v)
        r = Rotation.from_rotvec(v)
        np.testing.assert_allclose(np.abs(q @ r.as_quat()), 1.0, atol=1e-13)   # same rotation (q and -q are one)
        v2 = po.so3_log(q)
        np.testing.assert_allclose(Rotation.from_rotvec(v2).as_matrix(), r.as_matrix(), atol=1e-12)
    x = synth.make_state(pos=(1.5, -2.0, 0.7), rot=Rotation.from_rotvec([0.3, -0.2, 1.1]).as_quat(),
                         offR=Rotation.from_rotvec([0.01, 0.02, -0.03]).as_quat(), offT=(0.04, 0.02, -0.03))
    pts = RNG.uniform(-50, 50, (2000, 3)).astype(np.float32)
    w = po.points_body_to_world(x, pts)
    R, RL = Rotation.from_quat(x[3:7]), Rotation.from_quat(x[7:11])
    ref = R.apply(RL.apply(pts.astype(np.float64)) + x[11:14]) + x[0:3]
    np.testing.assert_allclose(w, ref.astype(np.float32), rtol=0, atol=8e-6)  # one fp32 rounding at |w| <= 100 m


def test_esti_plane_matches_numpy_least_squares():
    """esti_plane (include/common_lib.h:225-257): A n = -1 by QR; the unit normal and offset against numpy.linalg.lstsq, the
    verdict against the threshold recomputed in fp64."""
    ok_seen = bad_seen = 0
    for _ in range(300):
        n0 = RNG.normal(size=3)
        n0 /= np.linalg.norm(n0)
        c = RNG.uniform(-20, 20, 3)
        u = np.cross(n0, RNG.normal(size=3))
        u /= np.linalg.norm(u)
        v = np.cross(n0, u)
        noise = RNG.choice([0.0, 0.01, 0.2])
        P = (c + RNG.uniform(-1, 1, (5, 1)) * u + RNG.uniform(-1, 1, (5, 1)) * v + RNG.normal(0, noise, (5, 1)) * n0).astype(np.float32)
        ok, pabcd = po.esti_plane(P, 0.1)
        sol = np.linalg.lstsq(P.astype(np.float64), -np.ones(5), rcond=None)[0]
        nn = np.linalg.norm(sol)
        ref = np.r_[sol / nn, 1.0 / nn]
        res = np.abs(P.astype(np.float64) @ ref[:3] + ref[3])
        if res.max() < 0.09 or res.max() > 0.11:   # away from the threshold the verdict must agree
            assert ok == (res.max() <= 0.1)
        if ok:
            np.testing.assert_allclose(pabcd, ref, rtol=0, atol=5e-4 * max(1.0, abs(ref[3])) + 2e-4)  # fp32 QR at |p| <= 20 m
            ok_seen += 1
        else:
            bad_seen += 1
    assert ok_seen > 50 and bad_seen > 50


def _voxel_filter_reference(map_xyz, add_xyz, ds):
    """Add_Points(points, downsample_on = true), stated with dictionaries: per voxel touched by a new point, the point nearest
    to the voxel centre among the map's and the new ones survives (new beats old and later beats earlier at equal distance; a
    voxel whose single old point stays nearest is left alone); survivors keep the map's order, new ones are appended in input order."""
    def vox(p):
        return tuple(np.floor(p.astype(np.float64) / ds).astype(np.int64))

    def dist(p, k):
        c = (np.array(k, np.float64) * ds + 0.5 * ds).astype(np.float32)
        d = p - c
        return np.float32(np.float32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])

    old = {}
    for i, p in enumerate(map_xyz):
        old.setdefault(vox(p), []).append(i)
    new = {}
    for j, p in enumerate(add_xyz):
        new.setdefault(vox(p), []).append(j)
    dead, keep_new = set(), []
    for k, js in new.items():
        best_j, best_d = None, None
        for j in js:
            d = dist(add_xyz[j], k)
            if best_d is None or d <= best_d:
                best_j, best_d = j, d
        olds = old.get(k, [])
        if olds:
            ds_old = [dist(map_xyz[i], k) for i in olds]
            bi = int(np.argmin(ds_old))  # first (lowest index) among equals
            if ds_old[bi] < best_d:
                if len(olds) == 1:
                    continue
                dead.update(i for i in olds if i != olds[bi])
                continue
            dead.update(olds)
        keep_new.append(best_j)
    keep_old = [i for i in range(len(map_xyz)) if i not in dead]
    return np.vstack([map_xyz[keep_old], add_xyz[sorted(keep_new)]]).astype(np.float32)


@pytest.mark.parametrize("ds", [0.5, 0.3])
def test_map_add_matches_a_dictionary_voxel_filter(ds):
    m = RNG.uniform(-6, 6, (4000, 3)).astype(np.float32)
    add = np.vstack([m[RNG.integers(0, len(m), 1500)] + RNG.normal(0, 0.15, (1500, 3)), RNG.uniform(-8, 8, (1500, 3)),
                     (RNG.integers(-12, 12, (400, 3)) / 4.0)]).astype(np.float32)   # the last block ties exactly
    got = po.map_add(m, add, True, ds)
    want = _voxel_filter_reference(m, add, ds)
    assert got.shape == want.shape
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


def test_map_incremental_decisions_match_a_numpy_restatement_on_scipy_neighbours():
    """map_incremental (src/laserMapping.cpp:427-474) written again in numpy straight from those lines, fed with scipy's nearest
    neighbours instead of the oracle's: same add / no-downsample / skip decision for every scan point whose neighbour set is not
    a tie.  (Nearest_Points come from the LAST search -- at the search state -- while the world point and the voxel centre use the
    posterior, exactly as in the node.)"""
    pr = synth.make_problem(60000, 5000, "avia", cfg=1)
    fsm = 0.5
    m = po.Map(pr.map_xyz)
    sc = po.Scan(pr.body, nthreads=4)
    x_search, x_post = pr.x_prior, pr.x_true
    sc.h_share_model(m, x_search, True, False)
    w_ref, c_ref = sc.map_incremental_classify(m, x_post, fsm, True)
    # ---- the independent side
    mp = pr.map_xyz.astype(np.float32)
    tree = cKDTree(mp.astype(np.float64))
    ws = po.points_body_to_world(x_search, pr.body).astype(np.float64)   # where the last search looked
    d6, i6 = tree.query(ws, k=6)
    tie = np.diff(d6 ** 2, axis=1).min(axis=1) <= 1e-5 * np.maximum(d6[:, 5] ** 2, 1e-6)
    near = mp[i6[:, :5]]                                                 # points_near[0..4], nearest first
    w = po.points_body_to_world(x_post, pr.body)                         # feats_down_world (float)
    np.testing.assert_array_equal(w.view(np.uint32), w_ref.view(np.uint32))
    mid = (np.floor(w.astype(np.float64) / fsm) * fsm + 0.5 * fsm).astype(np.float32)

    def calc_dist(a, b):                                                 # include/common_lib.h: float arithmetic, x, y, z in turn
        d = (a - b).astype(np.float32)
        return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(np.float32) + d[..., 2] * d[..., 2]

    dist = calc_dist(w, mid)
    far = np.all(np.abs(near[:, 0].astype(np.float64) - mid.astype(np.float64)) > 0.5 * fsm, axis=1)
    veto = np.any(calc_dist(near, mid[:, None, :]) < dist[:, None], axis=1)
    cls = np.where(far, 2, np.where(veto, 0, 1)).astype(np.uint8)
    ok = ~tie
    assert ok.mean() > 0.95 and len(set(cls[ok])) == 3                   # all three classes occur
    np.testing.assert_array_equal(cls[ok], c_ref[ok])


@pytest.mark.parametrize("ext", [False, True])
def test_h_share_model_matches_a_float64_numpy_restatement(ext):
    """h_share_model (src/laserMapping.cpp:640-756) in numpy / scipy, fp64 throughout: scipy neighbours, lstsq planes, the
    selection rule, the Jacobian rows.  The oracle works in the reference's mixed fp32 / fp64, so flags are compared away from
    their thresholds and numbers to fp32 accuracy."""
    pr = synth.make_problem(60000, 4000, "avia", cfg=1)
    m = po.Map(pr.map_xyz)
    sc = po.Scan(pr.body, nthreads=4)
    x = pr.x_true
    assert sc.h_share_model(m, x, True, ext)
    sel = sc.selected.astype(bool)
    # ---- independent side
    R, RL, t, tL = Rotation.from_quat(x[3:7]), Rotation.from_quat(x[7:11]), x[0:3], x[11:14]
    pb = pr.body.astype(np.float64)
    pw = R.apply(RL.apply(pb) + tL) + t
    mp = pr.map_xyz.astype(np.float32).astype(np.float64)
    d, idx = cKDTree(mp).query(pw, k=5)
    gate = d[:, 4] ** 2 <= 5.0                                                             # :671
    P = mp[idx]                                                                            # (N, 5, 3)
    sol = np.stack([np.linalg.lstsq(P[i], -np.ones(5), rcond=None)[0] for i in range(len(pb))])
    nn = np.linalg.norm(sol, axis=1)
    nrm, dd = sol / nn[:, None], 1.0 / nn
    fit_res = np.abs(np.einsum("nkj,nj->nk", P, nrm) + dd[:, None]).max(axis=1)           # esti_plane's 0.1 threshold
    pd2 = np.einsum("nj,nj->n", nrm, pw) + dd
    s = 1 - 0.9 * np.abs(pd2) / np.sqrt(np.linalg.norm(pb, axis=1))                        # :693
    want = gate & (fit_res <= 0.1) & (s > 0.9)
    clear = (np.abs(d[:, 4] ** 2 - 5.0) > 1e-3) & (np.abs(fit_res - 0.1) > 2e-3) & (np.abs(s - 0.9) > 2e-3)
    assert clear.mean() > 0.8
    np.testing.assert_array_equal(sel[clear], want[clear])
    both = sel & want & clear
    assert both.sum() > 500 and sc.n_eff == int(sel.sum())
    # normals, residuals (normvec.intensity = pd2) and Jacobian rows of the selected points, in the oracle's row order
    nv = sc.normvec
    np.testing.assert_allclose(nv[both, :3], nrm[both], atol=2e-3)
    np.testing.assert_allclose(nv[both, 3], pd2[both], atol=2e-3)
    rows = np.cumsum(sel) - 1
    hx, h = sc.h_x, sc.h
    pthis = RL.apply(pb) + tL
    C_ = R.inv().apply(nv[:, :3].astype(np.float64))                                       # built from the oracle's own normals: tests the algebra
    A_ = np.cross(pthis, C_)
    B_ = np.cross(pb, RL.inv().apply(C_))
    k = np.flatnonzero(both)[:400]
    for i in k:
        r = hx[rows[i]]
        np.testing.assert_allclose(r[0:3], nv[i, :3], atol=1e-7)
        np.testing.assert_allclose(r[3:6], A_[i], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(r[6:9], B_[i] if ext else 0.0, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(r[9:12], C_[i] if ext else 0.0, rtol=1e-9, atol=1e-9)
        assert h[rows[i]] == -np.float64(nv[i, 3])


def test_undistortion_matches_the_reference_loop_written_out_in_python():
    """UndistortPcl's backward sweep (src/IMU_Processing.hpp:307-349) as the literal double loop over IMU segments and
    time-sorted points, scipy rotations, fp64 with the point stored back as float after every compensation (the cloud's fields
    are float) -- including the loop's exit through `if (it_pcl == begin) break`, which leaves the iterator ON the earliest point
    so that every older segment compensates it again.  Random poses, random extrinsic, 3 000 points."""
    rng = np.random.default_rng(5)
    n_pose, T = 9, 0.1
    ts = np.linspace(0.0, T, n_pose)
    rows, Rk = [], Rotation.identity()
    pos, vel = np.zeros(3), rng.normal(0, 2, 3)
    for k in range(n_pose):
        acc, gyr = rng.normal(0, 3, 3), rng.normal(0, 1.0, 3)
        rows.append((ts[k], acc, gyr, vel.copy(), pos.copy(), Rk.as_matrix().reshape(9)))
        dtk = T / (n_pose - 1)
        pos, vel, Rk = pos + vel * dtk, vel + acc * dtk, Rk * Rotation.from_rotvec(gyr * dtk)
    poses = po.make_poses(rows)
    x_end = synth.make_state(pos=pos + rng.normal(0, 0.01, 3), rot=Rk.as_quat(), offR=Rotation.from_rotvec([0.02, -0.01, 0.03]).as_quat(),
                             offT=(0.05, -0.02, 0.04))
    n = 3000
    tms = np.sort(rng.uniform(2.3 * T / (n_pose - 1) * 1000, T * 1000, n)).astype(np.float32)   # earliest point younger than IMUpose[2]
    pts = np.c_[rng.uniform(-60, 60, (n, 3)), tms].astype(np.float32)
    got = po.undistort(poses, x_end, pts)
    # ---- the loop
    out = pts[:, :3].copy()                                   # float fields
    R_end, R_LI = Rotation.from_quat(x_end[3:7]), Rotation.from_quat(x_end[7:11])
    p_end, T_LI = x_end[0:3], x_end[11:14]
    it = n - 1
    for kp in range(n_pose - 1, 0, -1):
        t_head, _, _, v_h, p_h, r_h = rows[kp - 1]
        acc_t, gyr_t = rows[kp][1], rows[kp][2]
        R_imu = Rotation.from_matrix(np.asarray(r_h).reshape(3, 3))
        while np.float64(pts[it, 3]) / 1000.0 > t_head:
            dt = np.float64(pts[it, 3]) / 1000.0 - t_head
            R_i = R_imu * Rotation.from_rotvec(np.asarray(gyr_t) * dt)
            P_i = out[it].astype(np.float64)
            T_ei = np.asarray(p_h) + np.asarray(v_h) * dt + 0.5 * np.asarray(acc_t) * dt * dt - p_end
            Pc = R_LI.inv().apply(R_end.inv().apply(R_i.apply(R_LI.apply(P_i) + T_LI) + T_ei) - T_LI)
            out[it] = Pc.astype(np.float32)
            if it == 0:
                break
            it -= 1
    np.testing.assert_allclose(got, out, rtol=0, atol=3e-5)    # fp32 storage of ~60 m coordinates: half an ulp is 2e-6; rotations differ at 1e-16
    # and the quirk is visible: the earliest point was carried again by the two segments older than its own
    once = po.undistort(poses, x_end, pts, first_point=False)
    assert np.abs(once[0] - got[0]).max() > 1e-3 and np.abs(once[1:] - got[1:]).max() == 0.0


@pytest.mark.parametrize("form", ["gain", "info"])
def test_iekf_pass_is_a_gauss_newton_step_on_the_manifold(form):
    """One pass of update_iterated_dyn_share_modified (IKFoM esekfom.hpp:1619-1931) is the Gauss-Newton step of
        min_d  || ((x [+] d) [-] x_prop) ||^2_{P^-1}  +  || H d - h ||^2 / R
    taken at a linearisation point x that is NOT the propagated state (so the manifold Jacobians of SO(3), SO(3) and S2 all
    matter).  Here the step is solved in numpy with the Jacobian of the prior term taken by central differences of
    boxminus(boxplus(.)) -- none of the closed-form A-matrix algebra of the reference -- and compared with the oracle's pass."""
    pr = synth.make_problem(60000, 3000, "avia", cfg=1)
    m = po.Map(pr.map_xyz)
    x_prop, P_prop = synth.propagate_prior_cov(lambda x, P, dt, Q, a, g: po.predict(x, P, dt, Q, a, g), pr.x_prior)
    rng = np.random.default_rng(3)
    d0 = np.r_[rng.normal(0, 0.02, 3), rng.normal(0, 0.01, 3), rng.normal(0, 0.005, 3), rng.normal(0, 0.01, 3), rng.normal(0, 0.05, 3),
               rng.normal(0, 1e-3, 6), rng.normal(0, 0.01, 2)]
    x = po.state_boxplus(x_prop, d0)
    sc = po.Scan(pr.body, nthreads=4)
    assert sc.h_share_model(m, x, True, True)
    H, h = sc.h_x, sc.h
    R = 0.001
    if form == "gain":
        x_new, P_new, Kx, dx = po.iekf_pass_gain(x, x_prop, P_prop, R, H[:20], h[:20])   # n_eff < 23: the gain-form branch
        H, h = H[:20], h[:20]
    else:
        x_new, P_new, Kx, dx = po.iekf_pass_info(x, x_prop, P_prop, R, H.T @ H, H.T @ h)
    # ---- numpy Gauss-Newton
    def e(d):
        return po.state_boxminus(po.state_boxplus(x, d), x_prop)
    eps = 1e-6
    J = np.stack([(e(eps * np.eye(23)[k]) - e(-eps * np.eye(23)[k])) / (2 * eps) for k in range(23)], axis=1)
    e0 = e(np.zeros(23))
    H23 = np.zeros((len(h), 23))
    H23[:, :12] = H
    Pi = np.linalg.inv(P_prop)
    A = J.T @ Pi @ J + H23.T @ H23 / R
    b = -J.T @ Pi @ e0 + H23.T @ h / R
    d = np.linalg.solve(A, b)
    scale = np.abs(d).max()
    # position, both rotations, lever arm, velocity, biases: the exact Gauss-Newton step
    np.testing.assert_allclose(dx[:21], d[:21], rtol=0, atol=1e-8 * max(scale, 1e-3))
    # gravity (S2): differs by ~1e-4 of the gravity correction, identically for every finite-difference step and for both
    # forms of the pass -- IKFoM's S2 Jacobian Mx takes `exp(Bu, scalar_type(1/2))` with 1/2 in integer arithmetic, i.e. the
    # identity (mtk S2.hpp:277; cf. tests/test_oracle_kat.py::test_S2_Mx_zero_delta_and_quirk), so it is the exact derivative
    # only at zero; the restatement keeps the reference's formulas, so this block is the reference's step, not the textbook one
    np.testing.assert_allclose(dx[21:], d[21:], rtol=2e-3, atol=1e-8)
    assert np.abs(dx[21:] - d[21:]).max() < 1e-5
    np.testing.assert_allclose(po.state_boxminus(x_new, po.state_boxplus(x, dx)), 0.0, atol=1e-9)


def test_predict_matches_scipy_propagation_and_a_finite_difference_covariance():
    """esekf::predict with FAST-LIO's process model (use-ikfom.hpp get_f / df_dx / df_dw, esekfom.hpp:280-381): the mean against
    the model written with scipy rotations (exact), the covariance against F P F^T + G Q G^T with F and G taken by central
    differences of that mean propagation.  The reference's F = F_x1 + f_x dt is first order in dt, so the two covariances differ
    at most in second order -- in fact not at all, every block being linear in dt -- with ONE exception the independent model
    exposes: the rotation's own block.  Its true Jacobian is Exp(-(w - bg) dt); esekfom.hpp:312 computes
    `exp(res.vec(), seg_SO3, scalar_type(1/2))`, and 1/2 is integer division, so the block stays the identity.  The restatement
    keeps that (oracle_math.c: "quirk, esekfom.hpp:312"), and this test pins it: with the finite-difference block replaced by I the
    two covariances agree to the noise of the differences; without the replacement they differ in first order (|w| dt)."""
    rng = np.random.default_rng(8)
    x = synth.make_state(pos=(1, -2, 0.5), rot=Rotation.from_rotvec([0.2, -0.4, 0.9]).as_quat(), offR=Rotation.from_rotvec([0.01, 0.0, 0.02]).as_quat(),
                         offT=(0.04, 0.02, -0.03), vel=(1.5, -0.7, 0.2), bg=(0.01, -0.02, 0.005))
    x[20:23] = (0.05, -0.03, 0.02)                                  # accelerometer bias
    A_ = rng.normal(size=(23, 23)) * 0.01
    P = A_ @ A_.T + np.diag(rng.uniform(1e-4, 1e-2, 23))
    Q = po.process_noise_cov()
    acc, gyr = np.array([0.3, -0.2, 9.9]), np.array([0.4, -0.1, 0.25])

    def run(dt):
        def mean(xx, a=acc, g=gyr):
            R = Rotation.from_quat(xx[3:7])
            y = xx.copy()
            y[0:3] = xx[0:3] + xx[14:17] * dt
            y[3:7] = (R * Rotation.from_rotvec((g - xx[17:20]) * dt)).as_quat()
            y[14:17] = xx[14:17] + (R.apply(a - xx[20:23]) + xx[23:26]) * dt
            return y

        x1, P1 = po.predict(x, P, dt, Q, acc, gyr)
        base = mean(x)
        np.testing.assert_allclose(po.state_boxminus(x1, base), 0.0, atol=1e-12)
        eps = 1e-6
        F = np.stack([(po.state_boxminus(mean(po.state_boxplus(x, eps * np.eye(23)[k])), base)
                       - po.state_boxminus(mean(po.state_boxplus(x, -eps * np.eye(23)[k])), base)) / (2 * eps) for k in range(23)], axis=1)

        # process noise (n_g, n_a, n_bg, n_ba): gyro and accelerometer noise through the measurement, the bias walks directly
        def mean_w(w):
            y = mean(x, acc - w[3:6], gyr - w[0:3])
            y[17:20] += w[6:9] * dt
            y[20:23] += w[9:12] * dt
            return y
        G = np.stack([(po.state_boxminus(mean_w(eps * np.eye(12)[k]), base) - po.state_boxminus(mean_w(-eps * np.eye(12)[k]), base)) / (2 * eps)
                      for k in range(12)], axis=1)
        true_rot_block = F[3:6, 3:6].copy()
        F[3:6, 3:6] = np.eye(3)                                        # esekfom.hpp:312, scalar_type(1/2) == 0
        err_quirk = np.abs(P1 - (F @ P @ F.T + G @ Q @ G.T)).max()
        F[3:6, 3:6] = true_rot_block
        err_true = np.abs(P1 - (F @ P @ F.T + G @ Q @ G.T)).max()
        return err_quirk, err_true, np.abs(P1 - P).max()

    e1, t1, step1 = run(0.005)
    e2, t2, _ = run(0.00125)
    assert e1 < 1e-10 and e2 < 1e-10 and step1 > 1e-5          # with the identity block: equal to the noise of the finite differences
    assert t1 > 1e-6 and 3.0 < t1 / t2 < 5.0                   # against the textbook Jacobian: off in FIRST order (4x for dt / 4)
