#!/usr/bin/env python
"""Developer tool (GPU box, ONE GPU): what the multi-GPU split of BASELINE configs[1] can gain, from measurements a single GPU
allows.

  1. the wall time of a searching / a no-search pass on ONE rank's share of the scan when the scan is sharded G ways (Morton-first
     shards, map replicated: G = 1, 2, 4, 8) -- a rank of a G-GPU run does exactly this work, so T(1) / T(G) bounds the strong scaling
     of the pass from above (the exchange comes on top);
  2. the exchange itself: the same pass with the group sums going out as peer-written granules (flh_peer_*; one rank, and two
     handles sharing this GPU through flh_eval_group) and through a one-rank RCCL communicator (ncclAllReduce + publish kernel).
Python-level timing (ctypes adds the same few microseconds to every variant): read the differences, not the absolutes."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast_lio_amd import capi, synth  # noqa: E402
from fast_lio_amd import dist as fdist  # noqa: E402

import argparse

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2, help="BASELINE config: 2 (100k vs 5M, sharded), 4 (130k Ouster vs 20M, sharded), "
                                                      "5 (200k MID-360 vs 50M, map PARTITIONED into slabs + halo, whole scan on every rank)")
ap.add_argument("--reps", type=int, default=150)
ap.add_argument("--only-shares", action="store_true", help="skip the exchange legs (peer granules, RCCL)")
args = ap.parse_args()
CFG = {2: (5_000_000, 100_000, "avia"), 4: (20_000_000, 130_000, "ouster64"), 5: (50_000_000, 200_000, "mid360")}
M, N, SENSOR = CFG[args.config]
REPS = args.reps
pr = synth.make_problem(M, N, SENSOR, cfg=args.config)
xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)


def timed(fn, reps=REPS):
    for _ in range(10):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6


def passes(ev):
    """(searching pass at the prior, searching pass at the true state, no-search pass) in us"""
    a = timed(lambda: ev(xp, True))
    b = timed(lambda: ev(pr.x_true, True))
    c = timed(lambda: ev(pr.x_true, False))
    return a, b, c


print(f"BASELINE configs[{args.config - 1}]: {N} scan points, {M} map points; {REPS} repetitions per figure; us per pass (search at the prior / "
      f"search at the true state / no search)")
base = None
if args.config == 5:
    # the map partitioned: a rank's work is the WHOLE scan against its slab (+ halo), fitting only the queries it owns; the pass
    # ends with the slowest rank, so every rank of a G-way partition is measured and the maximum reported
    for G in (1, 2, 4, 8):
        axis, edges = fdist.partition_bounds(pr.map_xyz, G)
        worst, sizes = None, []
        for r in range(G):
            keep = fdist.partition_slab(pr.map_xyz, axis, edges, r, fdist.HALO_DEFAULT)
            h = capi.Handle()
            h.map_build(pr.map_xyz[keep])
            if G > 1:
                h.set_owned_interval(axis, edges[r], edges[r + 1])
            h.scan_upload(pr.body)
            h.set_timing_stride(0)
            t = passes(lambda x, s: h.eval(x, s, False))
            sizes.append(int(keep.sum()) if keep.dtype == bool else len(keep))
            worst = t if worst is None else tuple(max(a, b) for a, b in zip(worst, t))
            h.close()
        if G == 1:
            base = worst
        print(f"  slowest rank of a {G}-way partition (slabs of {min(sizes)}..{max(sizes)} map points, whole scan): "
              f"{worst[0]:6.1f} / {worst[1]:6.1f} / {worst[2]:6.1f}   => pass speed-up bound {base[0] / worst[0]:.2f} / "
              f"{base[1] / worst[1]:.2f} / {base[2] / worst[2]:.2f}")
    sys.exit(0)
for G in (1, 2, 4, 8):
    idx = fdist.morton_shard(pr.body, 0, G)
    shard = np.ascontiguousarray(pr.body[idx])
    h = capi.Handle()
    h.map_build(pr.map_xyz)
    h.scan_upload(shard)
    h.set_timing_stride(0)
    t = passes(lambda x, s: h.eval(x, s, False))
    if G == 1:
        base = t
    print(f"  one rank's share of a {G}-way shard ({len(idx):6d} points): {t[0]:6.1f} / {t[1]:6.1f} / {t[2]:6.1f}"
          f"   => pass speed-up bound {base[0] / t[0]:.2f} / {base[1] / t[1]:.2f} / {base[2] / t[2]:.2f}")
    h.close()

if args.only_shares:
    sys.exit(0)
# the exchange
h = capi.Handle()
capi.peer_init_all([h])
h.map_build(pr.map_xyz)
h.scan_upload(pr.body)
h.set_timing_stride(0)
t = passes(lambda x, s: capi.eval_group([h], x, s, False))
print(f"  peer granules, one rank (whole scan):            {t[0]:6.1f} / {t[1]:6.1f} / {t[2]:6.1f}")
h.close()
hs = [capi.Handle(), capi.Handle()]
capi.peer_init_all(hs)
for r, hh in enumerate(hs):
    hh.map_build(pr.map_xyz)
    hh.scan_upload(np.ascontiguousarray(pr.body[fdist.morton_shard(pr.body, r, 2)]))
    hh.set_timing_stride(0)
t = passes(lambda x, s: capi.eval_group(hs, x, s, False))
print(f"  peer granules, two handles sharing this GPU:     {t[0]:6.1f} / {t[1]:6.1f} / {t[2]:6.1f}")
for hh in hs:
    hh.close()
try:
    h = capi.Handle()
    h.rccl_init_rank(1, capi.rccl_unique_id(), 0)
    h.map_build(pr.map_xyz)
    h.scan_upload(pr.body)
    h.set_timing_stride(0)
    t = passes(lambda x, s: h.eval(x, s, False))
    print(f"  RCCL all-reduce + publish kernel, one rank:      {t[0]:6.1f} / {t[1]:6.1f} / {t[2]:6.1f}   (one-launch searching pass, the group totals "
          f"all-reduced: {h.pass_stats()['one_launch_passes']} of {h.pass_stats()['search_passes']} searching passes in one launch)")
    h.close()
except Exception as e:  # noqa: BLE001
    print("  RCCL:", repr(e)[:200])
