"""Full-size GPU checks (BASELINE.json configs): the headline config against the oracle, the larger ones
through size-independent properties (linearity of the normal equations over scan shards, permutation
invariance, idempotence, run-to-run determinism, sortedness / gate of the neighbour lists)."""
import numpy as np
import pytest

from fast_lio_amd import capi, synth

pytestmark = pytest.mark.gpu


def _props(h, body, x, ext=False):
    N = len(body)
    h.scan_upload(body)
    HTH, HTh, n_eff, tres = h.eval(x, True, ext)
    sel = h.fetch_selected()
    idx, d2, cnt = h.fetch_neighbors()
    assert n_eff == int(sel.sum())
    # sortedness + gate: selected => five neighbours, ascending, 5th within sqrt(5) m; indices valid and distinct
    s = sel.astype(bool)
    assert np.all(cnt[s] == 5) and np.all(np.diff(d2[s], axis=1) >= 0) and np.all(d2[s][:, 4] <= 5.0)
    assert idx[s].min() >= 0 and idx[s].max() < h.M
    srt = np.sort(idx[s], axis=1)
    assert np.all(srt[:, 1:] != srt[:, :-1])
    np.testing.assert_allclose(HTH, HTH.T, rtol=0, atol=1e-12 * np.abs(HTH).max())
    assert np.all(np.linalg.eigvalsh(HTH[:6, :6]) > -1e-9 * np.abs(HTH).max())
    # idempotence: a no-search pass and a second search at the same state change nothing
    for search in (False, True):
        H2, h2, n2, t2 = h.eval(x, search, ext)
        np.testing.assert_array_equal(H2, HTH)
        np.testing.assert_array_equal(h2, HTh)
        assert n2 == n_eff
        np.testing.assert_array_equal(h.fetch_selected(), sel)
    # linearity over shards (what the multi-GPU all-reduce relies on): Gram(all) = Gram(first half) + Gram(second)
    half = N // 2
    parts = []
    for lo, hi in ((0, half), (half, N)):
        h.scan_upload(body[lo:hi])
        parts.append(h.eval(x, True, ext))
    np.testing.assert_allclose(parts[0][0] + parts[1][0], HTH, rtol=0, atol=1e-11 * np.abs(HTH).max())
    np.testing.assert_allclose(parts[0][1] + parts[1][1], HTh, rtol=0, atol=1e-11 * max(np.abs(HTh).max(), 1e-30) + 1e-9)
    assert parts[0][2] + parts[1][2] == n_eff
    # permutation invariance: flags follow the points, the normal equations do not move
    perm = np.random.default_rng(5).permutation(N)
    h.scan_upload(body[perm])
    H3, h3, n3, _ = h.eval(x, True, ext)
    np.testing.assert_array_equal(h.fetch_selected(), sel[perm])
    np.testing.assert_array_equal(h.fetch_neighbors()[0][sel[perm].astype(bool)], idx[perm][sel[perm].astype(bool)])
    np.testing.assert_allclose(H3, HTH, rtol=0, atol=1e-11 * np.abs(HTH).max())
    assert n3 == n_eff
    return n_eff


def test_config2_full_update_against_oracle():
    from oracle import pyoracle as po

    pr = synth.make_problem(5_000_000, 100_000, "avia", cfg=2)
    xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    h.scan_upload(pr.body)
    kf = capi.Esekf(h, max_iter=3)
    kf.change_x(xp)
    kf.change_P(P)
    st = kf.update(0.001)
    m = po.Map(pr.map_xyz)
    sc = po.Scan(pr.body, nthreads=16)
    x_ref, P_ref, st_ref = sc.update_iterated(m, xp, P)
    assert st.passes == st_ref.passes and st.searches == st_ref.searches
    assert list(st.n_eff)[: st.passes] == list(st_ref.n_eff)[: st_ref.passes]
    np.testing.assert_array_equal(h.fetch_selected(), sc.selected)          # bit-exact point_selected_surf
    x = kf.get_x()
    assert np.linalg.norm(x[:3] - x_ref[:3]) <= 1e-4                          # pose within 1e-4 m
    np.testing.assert_allclose(x, x_ref, rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(kf.get_P(), P_ref, rtol=0, atol=1e-4 * np.abs(P_ref).max())
    idx, d2, cnt = h.fetch_neighbors()
    gate = (sc.nn_cnt == 5) & (sc.nn_d2[:, 4] <= 5.0)
    np.testing.assert_array_equal(idx[gate], sc.nn_idx[gate])
    assert _props(h, pr.body, xp) > 50_000
    h.close()


def test_config4_properties_20M_map_130k_ouster():
    pr = synth.make_problem(20_000_000, 130_000, "ouster64", cfg=4)
    xp, _ = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    assert h.M == 20_000_000
    assert _props(h, pr.body, xp) > 30_000
    h.close()
