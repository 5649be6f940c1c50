"""One process per rank, granules written into every rank's pinned buffer (flh_peer_open): the exchange bench.py --gpus N uses.
Two processes share the one GPU of the box.  Every rank must end with the SAME posterior, bit for bit (each host adds the same
granules in the same order), and that posterior must be the single-process one up to the order of the fp64 sums."""
import os
import subprocess
import sys

import numpy as np
import pytest

from fast_lio_amd import capi, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("stale", [False, True])
def test_two_processes_exchange_granules(tmp_path, stale):
    """stale: a crashed earlier run has left a segment of the right size under the name, and rank 1 starts two seconds before
    rank 0 -- it maps the stale segment first; the attach handshake (nobody echoes its token there) sends it back to the name
    until rank 0 has replaced the segment (it used to sit in the stale one until both ranks timed out)."""
    world = 2
    name = f"/flh_peers_test_{os.getpid()}_{int(stale)}"
    outs = [str(tmp_path / f"r{r}.npz") for r in range(world)]
    worker = os.path.join(HERE, "_peer_worker.py")
    if stale:
        subprocess.run([sys.executable, worker, "0", str(world), name, str(tmp_path / "none.npz"), "crash"], timeout=300,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert os.path.exists("/dev/shm" + name), "the crashed run was meant to leave its segment behind"
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), name, outs[r]] + (["2.0"] if stale and r == 0 else []),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r}:\n{logs[r][-3000:]}"
    res = [np.load(o) for o in outs]
    pr = synth.make_problem(200000, 20000, "avia", cfg=1)
    xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    for tag, body, Pt in (("full", pr.body, P), ("few", pr.body[:20], P * 1e-6)):
        for k in ("_x", "_P", "_neff"):
            np.testing.assert_array_equal(res[0][tag + k], res[1][tag + k], err_msg=tag + k)
        h.scan_upload(np.ascontiguousarray(body))
        kf = capi.Esekf(h, max_iter=3)
        kf.change_x(xp)
        kf.change_P(Pt)
        st = kf.update(0.001)
        assert list(st.n_eff)[: st.passes] == list(res[0][tag + "_neff"]), tag
        np.testing.assert_allclose(kf.get_x(), res[0][tag + "_x"], rtol=0, atol=1e-10, err_msg=tag)
        # (the sums of the shards are added in another order than the single process adds its units: the posterior covariance
        # moves by that rounding times the conditioning of the information matrix; the contract is 1e-4 of max|P|)
        np.testing.assert_allclose(kf.get_P(), res[0][tag + "_P"], rtol=0, atol=1e-8 * np.abs(kf.get_P()).max(), err_msg=tag)
        sel = h.fetch_selected()
        for r in range(world):
            np.testing.assert_array_equal(res[r][tag + "_sel"], sel[res[r][tag + "_idx"]], err_msg=f"{tag}: flags of rank {r}")
        if tag == "few":
            assert min(st.n_eff[: st.passes]) < 23, "the case was meant to take the gain-form branch (gathered rows)"
        kf.close()
    h.close()
