"""The roofline events of bench.py: with a timing stride >= 2 a timed evaluation records its three HIP events and does not wait
for them; the elapsed times are read when the counters are asked for.  The sampling must not change any result, must count
exactly the evaluations it was asked to, and must survive more pending samples than the event pool holds."""
import numpy as np
import pytest

from fast_lio_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    pr = synth.make_problem(200000, 20000, "avia", cfg=1)
    xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
    h = capi.Handle(pass_kernel=0)  # the three-launch pass: search and fit are kernels of their own, each with its time stamps
    h.map_build(pr.map_xyz)
    h.scan_upload(pr.body)
    return pr, xp, h


def test_one_launch_pass_is_timed_as_one_kernel():
    """flh_config.pass_kernel (default): a searching evaluation is ONE kernel; its time is reported as the evaluation's and the
    search's, no fit sample is counted; no-search evaluations are timed as before."""
    pr = synth.make_problem(200000, 20000, "avia", cfg=1)
    xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    h.scan_upload(pr.body)
    ref_s = h.eval(xp, True, False)
    ref_n = h.eval(xp, False, False)
    t = h.timing()
    assert t["fit_ms"] > 0.0
    h.set_timing_stride(2)
    h.counters(reset=True)
    for k in range(12):  # sampled: k = 0, 2, ... (all searching)
        o = h.eval(xp, k % 2 == 0, False)
        r = ref_s if k % 2 == 0 else ref_n
        np.testing.assert_array_equal(o[0], r[0])
        assert o[2] == r[2]
    c = h.counters(reset=True)
    assert c["n_search"] == 6 and c["n_fit"] == 0 and c["n_eval"] == 6
    assert 0.0 < c["search_ms"] / c["n_search"] < 20.0 and c["eval_ms"] == c["search_ms"]
    h.set_timing_stride(2)
    for k in range(1, 13):  # sampled: k = 1, 3, ... -> wait: the count restarts at the stride change, so the 1st, 3rd, ... call
        h.eval(xp, k % 2 == 0, False)
    c = h.counters()
    assert c["n_search"] == 0 and c["n_fit"] == 6
    st = h.pass_stats()
    assert st["search_passes"] == st["one_launch_passes"] and st["search_passes"] >= 13
    h.close()


def test_deferred_event_reading_counts_and_leaves_results_alone(setup):
    pr, xp, h = setup
    h.set_timing_stride(0)
    ref_s = h.eval(xp, True, False)
    ref_n = h.eval(xp, False, False)
    h.set_timing_stride(2)
    h.counters(reset=True)
    outs = []
    for k in range(12):  # evaluations 0, 2, 4, ... are sampled; even k search, odd k do not
        outs.append(h.eval(xp, k % 2 == 0, False))
    c = h.counters()
    assert c["n_fit"] == 6 and c["n_search"] == 6   # every sampled evaluation was a searching one
    assert 0.0 < c["fit_ms"] / c["n_fit"] < 5.0 and 0.0 < c["search_ms"] / c["n_search"] < 20.0
    for k, o in enumerate(outs):
        r = ref_s if k % 2 == 0 else ref_n
        np.testing.assert_array_equal(o[0], r[0])
        np.testing.assert_array_equal(o[1], r[1])
        assert o[2] == r[2] and o[3] == r[3]
    t = h.timing()
    assert t["fit_ms"] > 0.0
    h.set_timing_stride(1)


def test_more_pending_samples_than_the_pool_holds(setup):
    pr, xp, h = setup
    h.set_timing_stride(0)
    h.eval(xp, True, False)  # the evaluations below reuse this search's neighbours
    h.set_timing_stride(2)
    h.counters(reset=True)
    n = 2 * 64 * 2 + 6  # 131 sampled evaluations: the pool of 64 triples is drained twice on the way
    for k in range(n):
        h.eval(xp, False, False)
    c = h.counters()
    assert c["n_fit"] == (n + 1) // 2 and c["n_search"] == 0
    assert 0.0 < c["fit_ms"] / c["n_fit"] < 5.0
    h.set_timing_stride(3)   # a change of the stride reads what is pending and restarts the count
    for k in range(7):
        h.eval(xp, k == 0, False)
    c = h.counters(reset=True)
    assert c["n_fit"] == (n + 1) // 2 + 3 and c["n_search"] == 1
    h.set_timing_stride(1)


def test_plane_cache_changes_no_bit():
    """flh_config.plane_cache (default on): no-search passes take each point's plane from the last searching pass instead of
    re-fitting it from the same five neighbours.  Every output must keep its bits: normal equations of a search / no-search /
    no-search sequence at three different states, flags, planes, and a whole iterated update."""
    pr = synth.make_problem(200000, 20000, "avia", cfg=1)
    xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
    x2 = np.array(xp, dtype=np.float64)
    x2[:3] += [0.02, -0.015, 0.01]
    x3 = np.array(pr.x_true, dtype=np.float64)
    got = []
    for on in (0, 1):
        h = capi.Handle(plane_cache=on)
        h.map_build(pr.map_xyz)
        h.scan_upload(pr.body)
        seq = []
        for ext in (False, True):
            for x, search in ((xp, True), (x2, False), (x3, False), (x3, True), (xp, False)):
                HTH, HTh, n_eff, tres = h.eval(x, search, ext)
                seq.append((HTH.copy(), HTh.copy(), n_eff, tres, h.fetch_selected().copy(), h.fetch_normvec().copy()))
        # flh_time_kernel re-runs the search WITHOUT a fit: the planes of the old neighbours must not be reused afterwards
        h.time_kernel(0, x3, False, 1)
        HTH, HTh, n_eff, tres = h.eval(x3, False, False)
        seq.append((HTH.copy(), HTh.copy(), n_eff, tres, h.fetch_selected().copy(), h.fetch_normvec().copy()))
        h.scan_upload(pr.body)
        kf = capi.Esekf(h, max_iter=3)
        kf.change_x(xp)
        kf.change_P(P)
        st = kf.update(0.001)
        got.append((seq, kf.get_x().copy(), kf.get_P().copy(), list(st.n_eff)[: st.passes]))
        kf.close()
        h.close()
    (seq0, x0, P0, n0), (seq1, x1, P1, n1) = got
    assert n0 == n1
    np.testing.assert_array_equal(x0, x1)
    np.testing.assert_array_equal(P0, P1)
    for a, b in zip(seq0, seq1):
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        assert a[2] == b[2] and a[3] == b[3]
        np.testing.assert_array_equal(a[4], b[4])
        sel = a[4].astype(bool)
        np.testing.assert_array_equal(a[5][sel].view(np.uint32), b[5][sel].view(np.uint32))


def test_search_only_sampling_counts_searching_evaluations(setup):
    """flh_set_timing_sampling(n, search_only=1): only searching evaluations are counted and timed, every n-th of them; a scan's
    first and later searches are told apart; results are untouched."""
    pr, xp, h = setup
    h.set_timing_stride(0)
    h.scan_upload(pr.body)
    ref_s = h.eval(xp, True, False)
    h.scan_upload(pr.body)
    h.set_timing_sampling(3, True)
    h.counters(reset=True)
    for k in range(14):  # searches at k = 0, 2, 4, ... (seven of them): the 1st, 4th and 7th are sampled
        o = h.eval(xp, k % 2 == 0, False)
        if k == 0:
            np.testing.assert_array_equal(o[0], ref_s[0])
    c = h.counters()
    sc = h.search_counters()
    assert c["n_search"] == 3 and c["n_fit"] == 3
    assert sc["n_first"] == 1 and sc["n_later"] == 2   # only the very first search of the scan is a "first search"
    assert 0.0 < c["search_ms"] / c["n_search"] < 20.0 and 0.0 < c["fit_ms"] / c["n_fit"] < 5.0
    h.set_timing_stride(1)


def test_one_launch_pass_equals_three_launch_pass():
    """flh_config.pass_kernel: the searching pass as ONE launch (k_pass: both search stages, fit, rows, Gram, group sums) against
    the three-launch pass -- flags, neighbour ids in rank order, distances, planes, the normal equations (both add the same
    64-point units in the same order: flh_fit_dev.hpp) and with them the whole update must be identical bit for bit -- on a dense
    scan, on a thinned-out scan (many queries reach the second stage), on a ragged size, on seven points, and with a prior so
    far off that most queries reach the second stage.  The one-launch pass with the neighbour cache kept as indices
    (flh_config.index_cache, the default) and as coordinates."""
    pr = synth.make_problem(200000, 20000, "avia", cfg=1)
    xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
    x_far = np.array(xp, dtype=np.float64)
    x_far[:3] += [1.6, -1.3, 1.7]          # metres off: 5th neighbours beyond the first stage's guaranteed radius
    scans = {"dense": pr.body, "sparse": np.ascontiguousarray(pr.body[::37]), "ragged": np.ascontiguousarray(pr.body[:5003]),
             "tiny": np.ascontiguousarray(pr.body[:7])}
    for name, body in scans.items():
        out = []
        variants = [dict(pass_kernel=0), dict(pass_kernel=1), dict(pass_kernel=1, index_cache=0)]
        for kw in variants:
            one = kw["pass_kernel"]
            h = capi.Handle(**kw)
            h.map_build(pr.map_xyz)
            h.scan_upload(body)
            res = []
            for ext in (False, True):
                for x, search in ((xp, True), (x_far, True), (pr.x_true, False), (pr.x_true, True), (xp, False)):
                    HTH, HTh, n_eff, tres = h.eval(x, search, ext)
                    idx, d2, cnt = h.fetch_neighbors()
                    res.append((HTH.copy(), HTh.copy(), n_eff, tres, h.fetch_selected().copy(), idx.copy(), cnt.copy(), d2.copy(),
                                h.fetch_normvec().copy()))
            st = h.pass_stats()
            assert st["one_launch_passes"] == (st["search_passes"] if one else 0), name
            if one and name == "dense":
                assert st["second_stage_queries"] > 0
            h.scan_upload(body)
            kf = capi.Esekf(h, max_iter=3)
            kf.change_x(xp)
            kf.change_P(P)
            us = kf.update(0.001)
            out.append((res, kf.get_x().copy(), kf.get_P().copy(), list(us.n_eff)[: us.passes]))
            kf.close()
            h.close()
        scan_name = name
        for vi in range(1, len(out)):
            (r0, x0, P0, n0), (r1, x1, P1, n1) = out[0], out[vi]
            name = f"{scan_name} {variants[vi]}"
            assert n0 == n1, name
            np.testing.assert_array_equal(x0, x1, err_msg=name)
            np.testing.assert_array_equal(P0, P1, err_msg=name)
            for a, b in zip(r0, r1):
                np.testing.assert_array_equal(a[4], b[4], err_msg=name + ": flags")
                np.testing.assert_array_equal(a[6], b[6], err_msg=name + ": neighbour counts")
                inside = a[7] <= 5.0                     # inside the gate the two must agree entry for entry
                np.testing.assert_array_equal(a[5][inside], b[5][inside], err_msg=name + ": neighbour ids")
                np.testing.assert_array_equal(a[7][inside].view(np.uint32), b[7][inside].view(np.uint32), err_msg=name + ": distances")
                sel = a[4].astype(bool)
                np.testing.assert_array_equal(a[8][sel].view(np.uint32), b[8][sel].view(np.uint32), err_msg=name + ": planes")
                np.testing.assert_array_equal(a[0], b[0], err_msg=name)
                np.testing.assert_array_equal(a[1], b[1], err_msg=name)
                assert a[2] == b[2] and a[3] == b[3], name


def test_synchronous_and_asynchronous_staging_do_not_share_scratch_unguarded():
    """The staging thread and the synchronous entry points use the same scratch buffers and the same copy stream.  Round 2's
    unexplained `Memory access fault by GPU` fits this picture: a synchronous staging that grows a scratch buffer (free + new
    allocation) while the worker's staging of another slot still reads it.  Every staging now holds one mutex; this test hammers
    the pair with growing sizes and checks every staged scan byte for byte."""
    pr = synth.make_problem(200000, 20000, "avia", cfg=1)
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    rng = np.random.default_rng(3)
    big = np.ascontiguousarray(np.tile(pr.body, (12, 1)) + rng.normal(0, 0.01, (12 * len(pr.body), 3)).astype(np.float32))
    assert len(big) >= 20000 + 7000 * 11 + 9000 + 9000 * 11
    for rep in range(12):
        n_async = 20000 + 7000 * rep          # both sizes grow: every round re-allocates some scratch buffer
        n_sync = 9000 + 9000 * rep
        a = np.ascontiguousarray(big[:n_async])
        b = np.ascontiguousarray(big[n_async: n_async + n_sync])
        h.scan_stage_async(0, a)
        h.scan_stage(1, b)                    # synchronous, while slot 0 may still be under way on the worker
        h.scan_upload(b[: 1000 + rep])        # the third path (its own slot, the same scratch)
        h.scan_wait(0)
        for slot, want in ((0, a), (1, b)):
            h.scan_activate(slot)
            got = h.fetch_scan()
            np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    h.close()
