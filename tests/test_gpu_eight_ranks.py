"""The 8-rank code paths of the multi-GPU design, executed on ONE GPU (sections, windows and kPeersMax are laid
out for eight ranks, nothing above two had ever run): eight handles in one process attached to each other through
flh_peer_init_all, evaluated through flh_eval_group.

* sharded (BASELINE configs[3]): one scan's points split Morton-first over eight handles, map replicated;
* partitioned (configs[4]): the map cut into eight slabs + halo, the whole scan on every handle, owned intervals.

The sum the ranks' normal equations meet in is the reference's H^T H / H^T h (esekfom.hpp:1784,1804).  Documented order of that
sum: every host adds the (rank, group) granules in rank order, groups in order -- so with ONE group per rank (shards of at most
4 096 points) the result must be the sequential sum of the eight single-handle evaluations BIT FOR BIT; with several groups per
rank it equals the single-handle result up to the order of the fp64 additions.  No 8-GPU run exists; this is the code executing."""
import numpy as np
import pytest

from fast_lio_amd import capi, synth
from fast_lio_amd import dist as fdist

pytestmark = pytest.mark.gpu

RANKS = 8


@pytest.fixture(scope="module")
def prob():
    pr = synth.make_problem(200000, 20000, "avia", cfg=1)
    xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
    return pr, xp, P


def seq_sum(parts):
    """((p0 + p1) + p2) + ... : the order in which a host adds the ranks' single-group sections."""
    tot = [parts[0][0].copy(), parts[0][1].copy(), parts[0][2], parts[0][3]]
    for p in parts[1:]:
        tot = [tot[0] + p[0], tot[1] + p[1], tot[2] + p[2], tot[3] + p[3]]
    return tot


@pytest.mark.parametrize("ext", [False, True])
def test_eight_handles_share_one_scan(prob, ext):
    pr, xp, P = prob
    body = pr.body                                    # 20 000 points: 2 500 per rank = 40 units = ONE group per rank
    shards = [fdist.morton_shard(body, r, RANKS) for r in range(RANKS)]
    np.testing.assert_array_equal(np.sort(np.concatenate(shards)), np.arange(len(body)))
    one = capi.Handle()
    one.map_build(pr.map_xyz)
    one.scan_upload(body)
    solo = []                                         # every shard on a handle of its own, no peers: what each rank contributes
    for idx in shards:
        s = capi.Handle()
        s.map_build(pr.map_xyz)
        s.scan_upload(np.ascontiguousarray(body[idx]))
        solo.append(s)
    hs = [capi.Handle() for _ in range(RANKS)]
    capi.peer_init_all(hs)
    assert all(h.peer_size() == RANKS for h in hs) and sorted(h.peer_rank() for h in hs) == list(range(RANKS))
    for h, idx in zip(hs, shards):
        h.map_build(pr.map_xyz)
        h.scan_upload(np.ascontiguousarray(body[idx]))
    for x, search in ((xp, True), (pr.x_true, False), (pr.x_true, True), (xp, False)):
        ref = one.eval(x, search, ext)
        parts = [s.eval(x, search, ext) for s in solo]
        want = seq_sum(parts)
        got = capi.eval_group(hs, x, search, ext)
        np.testing.assert_array_equal(got[0], want[0])   # the documented order, bit for bit
        np.testing.assert_array_equal(got[1], want[1])
        assert got[2] == want[2] == ref[2] and got[3] == want[3]
        np.testing.assert_allclose(got[0], ref[0], rtol=0, atol=1e-12 * np.abs(ref[0]).max())
        np.testing.assert_allclose(got[1], ref[1], rtol=0, atol=1e-12 * np.abs(ref[1]).max())
        sel = one.fetch_selected()
        for h, idx in zip(hs, shards):
            np.testing.assert_array_equal(h.fetch_selected(), sel[idx])
    for h in hs + solo + [one]:
        h.close()


def test_eight_handles_several_groups_per_rank_and_the_whole_update(prob):
    """80 000 points: 10 000 per rank = 157 units = three groups per rank; then the filter over the group's sum."""
    pr, xp, P = prob
    rng = np.random.default_rng(2)
    body = np.ascontiguousarray(np.tile(pr.body, (4, 1)) + rng.normal(0, 0.02, (4 * len(pr.body), 3)).astype(np.float32))
    shards = [fdist.morton_shard(body, r, RANKS) for r in range(RANKS)]
    one = capi.Handle()
    one.map_build(pr.map_xyz)
    one.scan_upload(body)
    hs = [capi.Handle() for _ in range(RANKS)]
    capi.peer_init_all(hs)
    for h, idx in zip(hs, shards):
        h.map_build(pr.map_xyz)
        h.scan_upload(np.ascontiguousarray(body[idx]))
    for x, search in ((xp, True), (pr.x_true, False)):
        ref = one.eval(x, search, False)
        got = capi.eval_group(hs, x, search, False)
        again = capi.eval_group(hs, x, search, False)
        np.testing.assert_array_equal(got[0], again[0])  # run to run: the same bits
        np.testing.assert_allclose(got[0], ref[0], rtol=0, atol=1e-12 * np.abs(ref[0]).max())
        np.testing.assert_allclose(got[1], ref[1], rtol=0, atol=1e-12 * np.abs(ref[1]).max())
        assert got[2] == ref[2] and abs(got[3] - ref[3]) <= 1e-12 * max(abs(ref[3]), 1.0)
    kf1 = capi.Esekf(one, max_iter=3)
    kf1.change_x(xp); kf1.change_P(P)
    st1 = kf1.update(0.001)

    def model(x, converge):
        HTH, HTh, n_eff, tres = capi.eval_group(hs, x, converge, False)
        if n_eff < 1:
            return {"valid": False, "n_eff": 0}
        return {"valid": True, "n_eff": n_eff, "HTH": HTH, "HTh": HTh, "total_residual": tres}

    kf = capi.Esekf(None, max_iter=3)
    kf.set_meas_model(model)
    kf.change_x(xp); kf.change_P(P)
    st = kf.update(0.001)
    assert st.passes == st1.passes and list(st.n_eff)[: st.passes] == list(st1.n_eff)[: st1.passes]
    np.testing.assert_allclose(kf.get_x(), kf1.get_x(), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(kf.get_P(), kf1.get_P(), rtol=0, atol=1e-8 * np.abs(kf1.get_P()).max())
    sel = one.fetch_selected()
    for h, idx in zip(hs, shards):
        np.testing.assert_array_equal(h.fetch_selected(), sel[idx])
    for h in hs + [one]:
        h.close()


def test_eight_slabs_of_the_map_with_owned_intervals(prob):
    pr, xp, P = prob
    body = pr.body
    one = capi.Handle()
    one.map_build(pr.map_xyz)
    one.scan_upload(body)
    axis, edges = fdist.partition_bounds(pr.map_xyz, RANKS)
    hs = [capi.Handle() for _ in range(RANKS)]
    capi.peer_init_all(hs)
    sizes = []
    for r, h in enumerate(hs):
        keep = fdist.partition_slab(pr.map_xyz, axis, edges, r, fdist.HALO_DEFAULT)
        h.map_build(pr.map_xyz[keep])
        h.set_owned_interval(axis, edges[r], edges[r + 1])
        h.scan_upload(body)                            # the whole scan on every rank
        sizes.append(h.M)
    assert max(sizes) < 0.5 * one.M
    for x, search in ((xp, True), (pr.x_true, False), (pr.x_true, True)):
        ref = one.eval(x, search, False)
        got = capi.eval_group(hs, x, search, False)
        np.testing.assert_allclose(got[0], ref[0], rtol=0, atol=1e-12 * np.abs(ref[0]).max())
        np.testing.assert_allclose(got[1], ref[1], rtol=0, atol=1e-12 * np.abs(ref[1]).max())
        assert got[2] == ref[2] and abs(got[3] - ref[3]) <= 1e-12 * max(abs(ref[3]), 1.0)
        sels = np.stack([h.fetch_selected() for h in hs])
        assert sels.sum(axis=0).max() <= 1             # every point has one owner
        np.testing.assert_array_equal(sels.max(axis=0), one.fetch_selected())
    for h in hs + [one]:
        h.close()
