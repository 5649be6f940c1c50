"""Known-answer tests that pin the CPU oracle (the reference has no golden vectors,
so these are authored from the math the cited reference lines implement).  CPU only."""
import numpy as np
import pytest

from oracle import pyoracle as po

RNG = np.random.default_rng(1234)


def rotvec_R(v):
    th = np.linalg.norm(v)
    K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * K @ K


def quat_R(q):
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


# ---------------------------------------------------------------- SO3 / A_matrix (mtkmath.hpp)
@pytest.mark.parametrize("scale", [1e-9, 1e-5, 1e-3, 0.1, 1.0, 2.5])
def test_so3_exp_log_roundtrip(scale):
    for _ in range(20):
        v = RNG.normal(size=3)
        v *= scale / np.linalg.norm(v)
        q = po.so3_exp(v)
        assert abs(np.linalg.norm(q) - 1) < 1e-14
        np.testing.assert_allclose(quat_R(q), rotvec_R(v), atol=1e-13)
        np.testing.assert_allclose(po.so3_log(q), v, rtol=1e-9, atol=1e-15)


def test_so3_exp_taylor_branch_matches_closed_form():
    # cos_sinc_sqrt switches to a 3-term Taylor pair below x^2 < eps^(1/4) (mtkmath.hpp:142-174)
    for n in [1e-3, 5e-3, 1e-2, 2e-2]:
        v = np.array([n, 0, 0])
        q = po.so3_exp(v)
        np.testing.assert_allclose(q, [np.sin(n / 2), 0, 0, np.cos(n / 2)], rtol=0, atol=1e-15)


def test_A_matrix_closed_form_and_identity():
    np.testing.assert_array_equal(po.A_matrix(np.zeros(3)), np.eye(3))
    np.testing.assert_array_equal(po.A_matrix(np.array([1e-12, 0, 0])), np.eye(3))
    for _ in range(10):
        v = RNG.normal(size=3)
        th = np.linalg.norm(v)
        K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
        ref = np.eye(3) + (1 - np.cos(th)) / th**2 * K + (1 - np.sin(th) / th) / th**2 * K @ K
        np.testing.assert_allclose(po.A_matrix(v), ref, atol=1e-14)


def test_A_matrix_is_right_jacobian_of_exp():
    # d/dw log(exp(v)^-1 exp(v + w)) at w=0 equals A(v)^T-like Jacobian: check A(v) w ~ log(exp(-v) exp(v+w))^T-form
    v = np.array([0.3, -0.2, 0.5])
    eps = 1e-7
    J = np.zeros((3, 3))
    for i in range(3):
        w = np.zeros(3)
        w[i] = eps
        qa = po.so3_exp(v)
        qb = po.so3_exp(v + w)
        qa_c = np.array([-qa[0], -qa[1], -qa[2], qa[3]])
        J[:, i] = po.so3_log(po.quat_mul(qa_c, qb)) / eps
    # right Jacobian of SO(3): Jr(v) = I - (1-cos)/th^2 K + (th - sin)/th^3 K^2 = A(v)^T
    np.testing.assert_allclose(J, po.A_matrix(v).T, atol=1e-6)


def test_quat_rot_matches_matrix():
    for _ in range(10):
        q = RNG.normal(size=4)
        q /= np.linalg.norm(q)
        v = RNG.normal(size=3)
        np.testing.assert_allclose(po.quat_rot(q, v), quat_R(q) @ v, atol=1e-14)


# ---------------------------------------------------------------- S2 (S2.hpp)
def random_grav():
    g = RNG.normal(size=3)
    return g / np.linalg.norm(g) * 9.809


def test_S2_Bx_is_tangent_basis():
    for _ in range(10):
        g = random_grav()
        Bx = po.S2_Bx(g)
        np.testing.assert_allclose(Bx.T @ g, 0, atol=1e-12)          # tangent to the sphere at g
        np.testing.assert_allclose(Bx.T @ Bx, np.eye(2), atol=1e-12)  # orthonormal


def test_S2_Bx_fallback_at_antipode():
    Bx = po.S2_Bx(np.array([-9.809, 0.0, 0.0]))
    np.testing.assert_array_equal(Bx, np.array([[0, 0], [0, -1], [1, 0]], float))


def test_S2_boxplus_boxminus_roundtrip():
    for _ in range(20):
        g = random_grav()
        d = RNG.normal(size=2) * 0.3
        g2 = po.S2_boxplus(g, d)
        assert abs(np.linalg.norm(g2) - 9.809) < 1e-12
        np.testing.assert_allclose(po.S2_boxminus(g2, g), d, atol=1e-10)
    np.testing.assert_array_equal(po.S2_boxminus(g, g), [0, 0])


def test_S2_Mx_zero_delta_and_quirk():
    g = random_grav()
    K = np.array([[0, -g[2], g[1]], [g[2], 0, -g[0]], [-g[1], g[0], 0]])
    Bx = po.S2_Bx(g)
    np.testing.assert_allclose(po.S2_Mx(g, np.zeros(2)), -K @ Bx, atol=1e-13)
    d = np.array([0.02, -0.01])
    # scalar(1/2) == 0 -> leading rotation is the identity (S2.hpp:277)
    ref = -K @ po.A_matrix(Bx @ d).T @ Bx
    np.testing.assert_allclose(po.S2_Mx(g, d), ref, atol=1e-13)
    np.testing.assert_allclose(po.S2_Nx_yy(g), Bx.T @ K / 9.809**2, atol=1e-14)


def test_Nx_Mx_is_identity_at_zero():
    # N(x,x) M(x,0) = d/dd ((x [+] d) [-] x) = I_2
    g = random_grav()
    np.testing.assert_allclose(po.S2_Nx_yy(g) @ po.S2_Mx(g, np.zeros(2)), np.eye(2), atol=1e-12)


# ---------------------------------------------------------------- compound state
def random_state():
    x = np.zeros(26)
    x[0:3] = RNG.normal(size=3)
    q = RNG.normal(size=4)
    x[3:7] = q / np.linalg.norm(q)
    q = RNG.normal(size=4)
    x[7:11] = q / np.linalg.norm(q)
    x[11:23] = RNG.normal(size=12)
    x[23:26] = random_grav()
    return x


def test_state_boxplus_boxminus_roundtrip():
    for _ in range(10):
        x = random_state()
        d = RNG.normal(size=23) * 0.1
        y = po.state_boxplus(x, d)
        np.testing.assert_allclose(po.state_boxminus(y, x), d, atol=1e-10)
    np.testing.assert_allclose(po.state_boxminus(x, x), 0, atol=1e-15)


# ---------------------------------------------------------------- inverse
def test_inverse_23():
    for cond in [1e2, 1e8]:
        U, _ = np.linalg.qr(RNG.normal(size=(23, 23)))
        s = np.logspace(0, np.log10(cond), 23)
        A = (U * s) @ U.T
        Ai = po.inverse(A)
        np.testing.assert_allclose(Ai @ A, np.eye(23), atol=1e-7 * cond / 1e2 if cond > 1e4 else 1e-10)
        np.testing.assert_allclose(Ai, np.linalg.inv(A), rtol=1e-6, atol=1e-9 * np.abs(Ai).max())


# ---------------------------------------------------------------- esti_plane (common_lib.h:225-257)
def test_qr_solve_matches_lstsq():
    for _ in range(200):
        A = RNG.normal(size=(5, 3)).astype(np.float32) * RNG.uniform(0.1, 100)
        b = -np.ones(5, np.float32)
        x = po.qr_solve_5x3(A, b)
        ref = np.linalg.lstsq(A.astype(np.float64), b.astype(np.float64), rcond=None)[0]
        np.testing.assert_allclose(x, ref, rtol=2e-4, atol=2e-6 * np.abs(ref).max())


def test_esti_plane_exact_planes():
    # z = 1  ->  n = (0,0,-1), d = 1  (A n = -1  =>  n_z * 1 = -1)
    pts = np.array([[0, 0, 1], [1, 0, 1], [0, 1, 1], [1, 1, 1], [0.5, 0.3, 1]], np.float32)
    ok, p = po.esti_plane(pts)
    assert ok
    np.testing.assert_allclose(p, [0, 0, -1, 1], atol=1e-6)
    # tilted plane x + y + z = 3 -> normal -(1,1,1)/sqrt3, d = 3/sqrt3
    base = RNG.uniform(-1, 1, size=(5, 2))
    pts = np.stack([base[:, 0], base[:, 1], 3 - base[:, 0] - base[:, 1]], axis=1).astype(np.float32)
    ok, p = po.esti_plane(pts)
    assert ok
    np.testing.assert_allclose(p, np.array([-1, -1, -1, 3]) / np.sqrt(3), atol=2e-6)


def test_esti_plane_outlier_threshold():
    def plane_with_outlier(dz):
        pts = np.array([[0, 0, 5], [1, 0, 5], [0, 1, 5], [1, 1, 5], [0.5, 0.5, 5 + dz]], np.float32)
        return po.esti_plane(pts)

    assert plane_with_outlier(0.05)[0]
    assert not plane_with_outlier(0.5)[0]


def test_esti_plane_degenerate_collinear():
    # collinear points: rank-deficient; the LS solution still exists (min-norm-ish); result must be finite
    # or rejected, never crash.  Through-origin planes (d=0) cannot be represented by A n = -1.
    pts = np.array([[t, 2 * t, 3 * t + 1] for t in range(5)], np.float32)
    ok, p = po.esti_plane(pts)
    assert np.all(np.isfinite(p)) or not ok


def test_esti_plane_far_from_origin_fp32():
    # absolute (un-centred) coordinates ~500 m: fp32 conditioning still yields mm-level planes
    for _ in range(50):
        c = RNG.uniform(-500, 500, 3)
        n = RNG.normal(size=3)
        n /= np.linalg.norm(n)
        u = np.cross(n, [1, 0, 0])
        u /= np.linalg.norm(u)
        v = np.cross(n, u)
        ab = RNG.uniform(-0.7, 0.7, size=(5, 2))
        pts = (c + ab[:, :1] * u + ab[:, 1:] * v).astype(np.float32)
        ok, p = po.esti_plane(pts)
        d_true = -n @ c
        if abs(d_true) < 5:  # near-origin planes are ill-posed for the A n = -1 parametrisation
            continue
        sgn = np.sign(p[:3] @ n)
        assert ok
        np.testing.assert_allclose(p[:3] * sgn, n, atol=5e-3)


# ---------------------------------------------------------------- kNN
def test_knn_kdtree_vs_brute_random_and_lattice():
    pts = RNG.uniform(-20, 20, size=(5000, 3)).astype(np.float32)
    m = po.Map(pts)
    for _ in range(200):
        q = RNG.uniform(-22, 22, 3).astype(np.float32)
        n1, i1, d1 = m.knn5(q)
        n2, i2, d2 = m.knn5_brute(q)
        assert n1 == n2 == 5
        np.testing.assert_array_equal(i1, i2)
        np.testing.assert_array_equal(d1, d2)
        assert np.all(np.diff(d1) >= 0)
    # tie-heavy lattice, duplicated points: ties resolved by lower map index
    g = np.arange(-5, 6, dtype=np.float32)
    lat = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    lat = np.concatenate([lat, lat[:100]], axis=0)
    m = po.Map(lat)
    for _ in range(200):
        q = (RNG.integers(-5, 6, 3) + RNG.choice([0.0, 0.5], 3)).astype(np.float32)
        n1, i1, d1 = m.knn5(q)
        n2, i2, d2 = m.knn5_brute(q)
        np.testing.assert_array_equal(i1, i2)
        np.testing.assert_array_equal(d1, d2)


def test_knn_fewer_than_five():
    pts = RNG.uniform(-1, 1, size=(3, 3)).astype(np.float32)
    m = po.Map(pts)
    n, idx, d2 = m.knn5(np.zeros(3, np.float32))
    assert n == 3 and np.all(idx[3:] == -1) and np.all(np.isinf(d2[3:]))
