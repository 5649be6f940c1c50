"""Full-size GPU checks (BASELINE.json configs[1..4]): every config against the oracle at its full size (flags,
neighbour indices inside the gate and planes bit-exact; posterior within 1e-4 relative, pose within 1e-4 m), plus
size-independent properties (linearity of the normal equations over scan shards, permutation invariance, idempotence,
run-to-run determinism, sortedness / gate of the neighbour lists)."""
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
    """BASELINE configs[1], the headline: the same helper as configs 4 and 5 -- flags, planes + pd2 (normvec), neighbour indices
    AND their squared distances bit for bit, posterior within the bars -- plus the size-independent properties; then the same
    with extrinsic_est_en = 1, the reference's default (src/laserMapping.cpp:789, config/horizon.yaml:20): the twelve-column rows
    and the 93-slot granule sections at full size."""
    assert _full_update_against_oracle(5_000_000, 100_000, "avia", 2, nthreads=16, props=True, exts=(False, True)) > 50_000


def test_config4_properties_20M_map_130k_ouster():
    pr = synth.make_problem(20_000_000, 130_000, "ouster64", cfg=4)
    xp, _ = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    assert h.M == 20_000_000
    assert _props(h, pr.body, xp) > 30_000
    h.close()

def _full_update_against_oracle(M, N, sensor, cfg, nthreads=32, props=True, exts=(False,)):
    from oracle import pyoracle as po

    pr = synth.make_problem(M, N, sensor, cfg=cfg)
    xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    assert h.M == M
    m = po.Map(pr.map_xyz)
    sc = po.Scan(pr.body, nthreads=nthreads)
    n_sel = 0
    for ext in exts:
        h.scan_upload(pr.body)
        kf = capi.Esekf(h, max_iter=3, extrinsic_est_en=ext)
        kf.change_x(xp)
        kf.change_P(P)
        st = kf.update(0.001)
        x_ref, P_ref, st_ref = sc.update_iterated(m, xp, P, extrinsic_est_en=ext)
        assert st.passes == st_ref.passes and st.searches == st_ref.searches
        assert list(st.n_eff)[: st.passes] == list(st_ref.n_eff)[: st_ref.passes]
        np.testing.assert_array_equal(h.fetch_selected(), sc.selected)          # bit-exact point_selected_surf
        sel = sc.selected.astype(bool)
        nv = h.fetch_normvec()[sel]
        if not ext:
            np.testing.assert_array_equal(nv.view(np.uint32), sc.normvec[sel].view(np.uint32))  # planes + pd2
        else:
            # The plane normals depend on the neighbours only: bit for bit.  pd2 is taken at the LAST pass's state, and with
            # extrinsic estimation the product's 12 x 12 information form and the reference's double inversion (the oracle's) leave
            # that state 1e-10 .. 1e-8 apart -- enough to move the fp32 world coordinate of a few points by an
            # ulp (8e-6 m at 100 m), i.e. pd2 -- a sum of four terms of that size -- by a few of them (measured on the MI355X: 0.2 %
            # of the points differ at all, by at most 3.1e-5 m).  The flags above were still identical.
            np.testing.assert_array_equal(nv[:, :3].view(np.uint32), sc.normvec[sel][:, :3].view(np.uint32))
            np.testing.assert_allclose(nv[:, 3], sc.normvec[sel][:, 3], rtol=0, atol=2e-4)
            assert (nv[:, 3].view(np.uint32) != sc.normvec[sel][:, 3].view(np.uint32)).mean() < 0.02
        x = kf.get_x()
        assert np.linalg.norm(x[:3] - x_ref[:3]) <= 1e-4                          # pose within 1e-4 m
        np.testing.assert_allclose(x, x_ref, rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(kf.get_P(), P_ref, rtol=0, atol=1e-4 * np.abs(P_ref).max())
        idx, d2, cnt = h.fetch_neighbors()
        gate = (sc.nn_cnt == 5) & (sc.nn_d2[:, 4] <= 5.0)
        np.testing.assert_array_equal(idx[gate], sc.nn_idx[gate])
        if not ext:
            np.testing.assert_array_equal(d2[gate].view(np.uint32), sc.nn_d2[gate].view(np.uint32))
        else:  # (pointSearchSqDis is taken at the last SEARCHING pass's state: same remark as for pd2 above)
            np.testing.assert_allclose(d2[gate], sc.nn_d2[gate], rtol=1e-4, atol=1e-5)
        n_sel = int(sel.sum())
        kf.close()
    if props:
        _props(h, pr.body, xp)
        if True in exts:
            _props(h, pr.body, xp, ext=True)
    h.close()
    return n_sel


def test_config4_full_update_against_oracle_20M_map_130k_ouster():
    """BASELINE configs[3]: Ouster-64 130k-point scan (beyond the reference's static 100000-point caps,
    src/laserMapping.cpp:76,94,112-114 -- the oracle lifts them) against the 20M-point map."""
    assert _full_update_against_oracle(20_000_000, 130_000, "ouster64", 4, props=False) > 30_000


def test_config5_full_update_against_oracle_50M_map_200k_mid360():
    """BASELINE configs[4]: MID-360 200k-point scan against the 50M-point map (the map exceeds the 256 MiB Infinity Cache
    by far; exercises the 11-bit packed-index path and the storage-size limits)."""
    assert _full_update_against_oracle(50_000_000, 200_000, "mid360", 5, props=False) > 30_000


def test_config3_velodyne_stream_10M_map_with_incremental_inserts_against_oracle():
    """BASELINE configs[2] at full size: a Velodyne scan stream against the 10M-point map, map_incremental after every
    update (src/laserMapping.cpp:879-927, five scans): posterior, flags and classes equal the oracle's on every scan and
    the device map stays bit-identical to the oracle's map."""
    from oracle import pyoracle as po

    M, N, DS = 10_000_000, 60_000, 0.5
    pr0 = synth.make_problem(M, N, "velodyne", cfg=3)
    scene = pr0.scene
    h = capi.Handle()
    h.map_build(pr0.map_xyz)
    kf = capi.Esekf(h, max_iter=3, extrinsic_est_en=False)
    cur = pr0.map_xyz.astype(np.float32)
    grown = 0
    for k in range(5):
        pr = synth.make_problem(M, N, "velodyne", cfg=3, scan_seed=k, scene=scene)
        xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
        m = po.Map(cur)
        h.scan_upload(pr.body)
        st = kf.update_scan(-1, np.ascontiguousarray(xp), np.ascontiguousarray(P), 0.001)
        sc = po.Scan(pr.body, nthreads=32)
        x_ref, P_ref, st_ref = sc.update_iterated(m, xp, P)
        assert st.passes == st_ref.passes and st.searches == st_ref.searches, f"scan {k}"
        assert list(st.n_eff)[: st.passes] == list(st_ref.n_eff)[: st.passes], f"scan {k}"
        np.testing.assert_array_equal(h.fetch_selected(), sc.selected, err_msg=f"scan {k}")
        idx, d2, cnt = h.fetch_neighbors()
        gate = (sc.nn_cnt == 5) & (sc.nn_d2[:, 4] <= 5.0)
        np.testing.assert_array_equal(idx[gate], sc.nn_idx[gate], err_msg=f"scan {k}")
        x_post = kf.get_x()
        assert np.linalg.norm(x_post[0:3] - x_ref[0:3]) <= 1e-4
        np.testing.assert_allclose(x_post, x_ref, rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(kf.get_P(), P_ref, rtol=0, atol=1e-4 * np.abs(P_ref).max())
        w_ref, c_ref = sc.map_incremental_classify(m, x_post, DS, True)
        h.map_incremental(x_post, DS, True, apply=True)
        w, c = h.fetch_map_incremental()
        np.testing.assert_array_equal(c, c_ref, err_msg=f"scan {k}")
        new = po.map_add(po.map_add(cur, w_ref[c_ref == 1], True, DS), w_ref[c_ref == 2], False, DS)
        grown += len(new) - len(cur)
        cur = new
        assert h.M == len(cur)
        if k in (0, 4):  # 120 MB each way: the first and the last scan
            got = h.map_download()
            assert got.shape == cur.shape
            np.testing.assert_array_equal(got.view(np.uint32), cur.view(np.uint32), err_msg=f"map after scan {k}")
    assert grown > 1000
    stats = h.map_stats()
    assert stats["brickwise"] >= 4  # the inserts went through the brick-wise path, not through full re-indexings
    h.close()
