"""Worker of tests/test_gpu_peers.py: one rank of a scan sharded over `world` processes that exchange their normal equations
as peer-written granules (flh_peer_open).  All ranks share GPU 0 here (the box has one); on a node each would take its own.
usage: python _peer_worker.py <rank> <world> <shm name> <out.npz> [delay_s | crash]
  delay_s: sleep that long before flh_peer_open (rank 0 of the stale-segment case comes late);
  crash:   be rank 0 of a run whose other ranks never come and die inside flh_peer_open -- what a crashed run leaves under the name."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fast_lio_amd import capi, synth  # noqa: E402
from fast_lio_amd import dist as fdist  # noqa: E402


def main():
    rank, world, name, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    extra = sys.argv[5] if len(sys.argv) > 5 else ""
    if extra == "crash":  # rank 0 of a run of `world` ranks that dies right after it has created the segment
        h = capi.Handle(device=0)
        L = capi.lib()
        # (the other ranks never come: flh_peer_open would wait for them, so the segment is left behind from inside the wait)
        import threading

        threading.Timer(1.5, lambda: os._exit(0)).start()
        L.flh_peer_open(h.ptr, name.encode(), world, 0)
        os._exit(0)
    if extra:
        time.sleep(float(extra))
    pr = synth.make_problem(200000, 20000, "avia", cfg=1)
    xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
    h = capi.Handle(device=0)
    h.peer_open(name, world, rank)
    h.map_build(pr.map_xyz)
    idx = fdist.morton_shard(pr.body, rank, world)
    res = {}
    for tag, body in (("full", pr.body), ("few", pr.body[:20])):  # "few": 20 points in all -> n_eff < 23: the gathered rows
        idx = fdist.morton_shard(body, rank, world)
        h.scan_upload(np.ascontiguousarray(body[idx]))
        kf = capi.Esekf(h, max_iter=3)
        kf.change_x(xp)
        kf.change_P(P if tag == "full" else P * 1e-6)
        st = kf.update(0.001)
        res[tag + "_x"] = kf.get_x().copy()
        res[tag + "_P"] = kf.get_P().copy()
        res[tag + "_neff"] = np.array(list(st.n_eff)[: st.passes])
        res[tag + "_sel"] = h.fetch_selected().copy()
        res[tag + "_idx"] = idx
        kf.close()
    np.savez(out, **res)
    h.close()


if __name__ == "__main__":
    main()
