"""The product's host-side IEKF (include/fastlio_amd/esekfom.hpp via the C ABI) against the oracle.
The measurement model is injected through flh_esekf_set_meas_model and backed by the oracle's
h_share_model, so these run without a GPU."""
import numpy as np
import pytest

from fast_lio_amd import capi, synth
from oracle import pyoracle as po


@pytest.fixture(scope="module")
def small():
    pr = synth.make_problem(60000, 4000, "avia", cfg=102)
    m = po.Map(pr.map_xyz)
    xp, P = synth.propagate_prior_cov(po.predict, pr.x_prior)
    return pr, m, xp, P


def oracle_model(sc, m, ext, rows=False, log=None):
    def fn(x, converge):
        valid = sc.h_share_model(m, x, converge, ext)
        if log is not None:
            log.append((converge, sc.n_eff))
        if not valid:
            return {"valid": False, "n_eff": 0}
        if rows:
            return {"valid": True, "n_eff": sc.n_eff, "h_x": sc.h_x, "h": sc.h, "total_residual": sc.total_residual}
        HTH, HTh = sc.normal_equations()
        return {"valid": True, "n_eff": sc.n_eff, "HTH": HTH, "HTh": HTh, "total_residual": sc.total_residual}

    return fn


def test_predict_matches_oracle(small):
    pr, m, xp, P = small
    Q = synth.process_noise_cov()
    acc = np.array([0.1, -0.2, 9.7])
    gyro = np.array([0.01, 0.02, -0.03])
    x1, P1 = po.predict(xp, P, 0.005, Q, acc, gyro)
    x2, P2 = capi.predict_fn(xp, P, 0.005, Q, acc, gyro)
    np.testing.assert_allclose(x2, x1, rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(P2, P1, rtol=1e-12, atol=1e-16)
    # the prior used everywhere: 10 predict steps from the IMU_init covariance
    _, Pa = synth.propagate_prior_cov(po.predict, pr.x_prior)
    _, Pb = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
    np.testing.assert_allclose(Pb, Pa, rtol=1e-11, atol=1e-18)


@pytest.mark.parametrize("ext", [False, True])
@pytest.mark.parametrize("rows", [False, True])
def test_update_matches_oracle(small, ext, rows):
    pr, m, xp, P = small
    ref = po.Scan(pr.body, nthreads=2)
    x_ref, P_ref, st_ref = ref.update_iterated(m, xp, P, extrinsic_est_en=ext)
    sc = po.Scan(pr.body, nthreads=2)
    log = []
    kf = capi.Esekf(None, max_iter=3, extrinsic_est_en=ext)
    kf.set_meas_model(oracle_model(sc, m, ext, rows=rows, log=log))
    kf.change_x(xp)
    kf.change_P(P)
    st = kf.update(0.001)
    assert st.passes == st_ref.passes and st.searches == st_ref.searches
    assert list(st.pass_search)[: st.passes] == list(st_ref.pass_search)[: st.passes]
    assert list(st.n_eff)[: st.passes] == list(st_ref.n_eff)[: st.passes]
    assert st.returned_in_loop == st_ref.returned_in_loop == 1
    x, Pn = kf.get_x(), kf.get_P()
    np.testing.assert_allclose(x, x_ref, rtol=1e-9, atol=1e-11)
    # P = L - K_x P cancels heavily for well-observed states (K H ~ I), so rounding-level differences in the
    # order of the 23x23 algebra show up as ~1e-7 * max|P| on small off-diagonal entries: compare norm-wise.
    np.testing.assert_allclose(Pn, P_ref, rtol=0, atol=1e-6 * np.abs(P_ref).max())


def test_gain_form_branch_small_n(small):
    # fewer than 23 effective rows -> esekfom.hpp:1715-1744 (needs explicit rows)
    pr, m, xp, P = small
    body = pr.body[:16]
    ref = po.Scan(body, nthreads=1)
    x_ref, P_ref, st_ref = ref.update_iterated(m, xp, P)
    assert 0 < max(st_ref.n_eff) < 23
    sc = po.Scan(body, nthreads=1)
    kf = capi.Esekf(None, max_iter=3)
    kf.set_meas_model(oracle_model(sc, m, False, rows=True))
    kf.change_x(xp)
    kf.change_P(P)
    st = kf.update(0.001)
    assert st.passes == st_ref.passes
    np.testing.assert_allclose(kf.get_x(), x_ref, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(kf.get_P(), P_ref, rtol=0, atol=1e-6 * np.abs(P_ref).max())


def test_invalid_measurement_skips_every_pass(small):
    pr, m, xp, P = small
    kf = capi.Esekf(None, max_iter=3)
    calls = []

    def fn(x, converge):
        calls.append(converge)
        return {"valid": False, "n_eff": 0}

    kf.set_meas_model(fn)
    kf.change_x(xp)
    kf.change_P(P)
    st = kf.update(0.001)
    assert st.passes == 4 and st.returned_in_loop == 0 and calls == [True] * 4
    np.testing.assert_array_equal(kf.get_x(), xp)
    np.testing.assert_array_equal(kf.get_P(), P)


def test_max_iter_variants(small):
    pr, m, xp, P = small
    for mi in (0, 1, 2, 5):
        ref = po.Scan(pr.body, nthreads=2)
        x_ref, P_ref, st_ref = ref.update_iterated(m, xp, P, max_iter=mi)
        sc = po.Scan(pr.body, nthreads=2)
        kf = capi.Esekf(None, max_iter=mi)
        kf.set_meas_model(oracle_model(sc, m, False))
        kf.change_x(xp)
        kf.change_P(P)
        st = kf.update(0.001)
        assert st.passes == st_ref.passes and st.returned_in_loop == st_ref.returned_in_loop
        np.testing.assert_allclose(kf.get_x(), x_ref, rtol=1e-9, atol=1e-11)


def test_update_scan_is_change_x_change_P_update(small):
    """flh_esekf_update_scan (one call per loop body) with slot < 0 = the three calls it replaces, bit for bit; and it
    refuses to activate a slot when the filter has no device handle."""
    pr, m, xp, P = small
    outs = []
    for one_call in (False, True):
        sc = po.Scan(pr.body, nthreads=2)
        kf = capi.Esekf(None, max_iter=3)
        kf.set_meas_model(oracle_model(sc, m, False))
        if one_call:
            st = kf.update_scan(-1, np.ascontiguousarray(xp, np.float64), np.ascontiguousarray(P, np.float64), 0.001)
        else:
            kf.change_x(xp)
            kf.change_P(P)
            st = kf.update(0.001)
        outs.append((kf.get_x(), kf.get_P(), st.passes, list(st.n_eff)[: st.passes]))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    assert outs[0][2:] == outs[1][2:]
    kf = capi.Esekf(None, max_iter=3)
    with pytest.raises(capi.FlhError):
        kf.update_scan(0, np.ascontiguousarray(xp, np.float64), np.ascontiguousarray(P, np.float64), 0.001)


def test_two_halves_measurement_model_changes_no_bit(tmp_path):
    """esekf with a measurement model whose first half is registered (the GPU pass: flh_eval_begin) does the covariance projection
    between the two halves; tests/cpp/split_model_check.cpp compares it with the one-piece flow bit for bit,
    including passes whose measurement is invalid."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "split_model_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "split_model_check.cpp"), "-o", str(exe)])
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode == 0 and "identical bits" in r.stdout.decode(), r.stdout.decode()
