#!/usr/bin/env python
"""Developer tool (GPU box): the staging kernels ALONE -- no update running beside them -- under rocprofv3, for scans in random order
(bench.py's synthetic scans) and in a spatially coherent order (what a real sensor delivers: tiles of consecutive records are compact
in space).   rocprofv3 --kernel-trace --stats ... -- python tools/stage_probe.py [--stage-sort 0|1] [--n 100000]
FLH_LIB selects another build of the library (fast_lio_amd/_build.py)."""
import argparse
import time

import numpy as np

from fast_lio_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--stage-sort", type=int, default=1)
ap.add_argument("--n", type=int, default=100000)
ap.add_argument("--reps", type=int, default=50)
args = ap.parse_args()
rng = np.random.default_rng(1)
# a plane patch + a wall, like a scan's footprint: ~0.5 m apart
g = int(np.sqrt(args.n)) + 1
xs, ys = np.meshgrid(np.arange(g) * 0.45 - 0.2 * g, np.arange(g) * 0.45 - 0.2 * g)
pts = np.stack([xs.ravel(), ys.ravel(), 0.02 * rng.standard_normal(g * g)], 1).astype(np.float32)[: args.n]
pts += rng.uniform(-0.1, 0.1, pts.shape).astype(np.float32)
clouds = {"coherent": np.ascontiguousarray(pts), "random": np.ascontiguousarray(pts[rng.permutation(len(pts))])}
h = capi.Handle(stage_sort=args.stage_sort)
h.map_build(pts[::7])
for name, body in clouds.items():
    pin = capi.pinned_empty(body.shape, np.float32)
    pin[:] = body
    h.scan_stage(0, pin)
    h.scan_activate(0)
    order = h.scan_order()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        h.scan_stage(1, pin)
    h.scan_activate(1)
    dt = (time.perf_counter() - t0) / args.reps
    assert np.array_equal(h.scan_order(), order)
    print(f"{name}: N={len(body)} stage_sort={args.stage_sort} {dt * 1e6:.1f} us per synchronous staging call (H2D + kernels + wait)")
h.close()
