#!/usr/bin/env python
"""GPU box: what does the pre-launched no-search pass (flh_eval_expect_next, flh_config.prelaunch) buy?  Parity is in
tests/test_gpu_parity.py::test_prelaunched_nosearch_pass_same_bits; this is the same-process A/B of the native scan loop, the
switch alternating, three times each:

    python tools/prelaunch_check.py [--M 5000000 --N 100000 --cfg 2 --steps 120]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast_lio_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=5_000_000)
ap.add_argument("--N", type=int, default=100_000)
ap.add_argument("--sensor", default="avia")
ap.add_argument("--cfg", type=int, default=2)
ap.add_argument("--scans", type=int, default=6)
ap.add_argument("--steps", type=int, default=120)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()

scene = synth.make_scene(args.M, synth.CONFIG_SEED_BASE + args.cfg)
probs = [synth.make_problem(args.M, args.N, args.sensor, cfg=args.cfg, scan_seed=k, scene=scene) for k in range(args.scans)]
priors = []
for pr in probs:
    xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
    priors.append((np.ascontiguousarray(xp, np.float64), np.ascontiguousarray(P, np.float64)))
h = capi.Handle()
h.map_build(probs[0].map_xyz)
h.set_timing_stride(0)  # no events: a timed evaluation is never handed to the mailbox
kf = capi.Esekf(h)
bodies = [np.ascontiguousarray(pr.body, np.float32) for pr in probs]
jobs = capi.Esekf.make_jobs(bodies, priors)
for rep in range(args.reps):
    for on in (0, 1):
        h.set_prelaunch(bool(on))
        c0 = h.prelaunch_stats()
        kf.run_scans(jobs, 0, 10)
        t0 = time.perf_counter()
        rs = kf.run_scans(jobs, 0, args.steps)
        dt = time.perf_counter() - t0
        c1 = h.prelaunch_stats()
        print(f"rep {rep} prelaunch {'on ' if on else 'off'}: {args.steps / dt:8.1f} scans/s  {1e3 * dt / args.steps:.4f} ms/scan  "
              f"searching pass {1e3 * rs.ms_search_passes / max(rs.n_search_passes, 1):.1f} us  "
              f"no-search pass {1e3 * rs.ms_nosearch_passes / max(rs.n_nosearch_passes, 1):.1f} us  passes/scan {rs.passes / rs.scans:.2f}  "
              f"mailbox {({k: c1[k] - c0[k] for k in c1})}")
kf.close()
h.close()
