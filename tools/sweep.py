#!/usr/bin/env python
"""Developer tool: time the hot path for several (lanes_per_query, cell_size, sort) variants in one process,
reusing one synthetic scene.  Prints one line per variant.  Needs a GPU."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast_lio_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--variants", default="4:1.5:1,8:1.5:1,4:1.0:1,0:1.5:1")
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--scans", type=int, default=4)
args = ap.parse_args()
from bench import CONFIGS  # noqa: E402

M, N, sensor = CONFIGS[args.config]
scene = synth.make_scene(M, synth.CONFIG_SEED_BASE + args.config)
probs = [synth.make_problem(M, N, sensor, cfg=args.config, scan_seed=s, scene=scene) for s in range(args.scans)]
priors = [synth.propagate_prior_cov(capi.predict_fn, p.x_prior) for p in probs]
for var in args.variants.split(","):
    lpq, cell, sort = var.split(":")
    h = capi.Handle(cell_size=float(cell), lanes_per_query=int(lpq), sort_queries=int(sort))
    h.map_build(scene.map_xyz)
    for s, p in enumerate(probs):
        h.scan_stage(s, p.body)
    kf = capi.Esekf(h, max_iter=3)
    h.set_timing_stride(int(os.environ.get('SWEEP_TIMING_STRIDE', '8')))

    def step(i):
        s = i % len(probs)
        h.scan_activate(s)
        kf.change_x(priors[s][0])
        kf.change_P(priors[s][1])
        return kf.update(0.001)

    for i in range(8):
        step(i)
    h.counters(reset=True)
    t0 = time.perf_counter()
    passes = 0
    hms = sms = 0.0
    for i in range(args.steps):
        st = step(i)
        passes += st.passes
        hms += st.h_ms
        sms += st.solve_ms
    dt = time.perf_counter() - t0
    c = h.counters()
    x0 = priors[0][0]
    h.scan_activate(0)
    ks = h.time_kernel(0, x0, False, 30) * 1e3
    kfit = h.time_kernel(1, x0, False, 30) * 1e3
    h.enable_stats(True)
    h.scan_activate(0)
    h.eval(x0, True, False)
    cand = h.timing()["candidates"] / N
    print(f"lpq={lpq:>2s} cell={cell} sort={sort}: {args.steps / dt:8.1f} scans/s  {dt / passes * 1e3:7.4f} ms/pass  "
          f"search(ev)={c['search_ms'] / max(c['n_search'], 1) * 1e3:7.2f}us fit(ev)={c['fit_ms'] / max(c['n_fit'], 1) * 1e3:6.2f}us  "
          f"host: h={hms / passes * 1e3:6.1f}us solve={sms / passes * 1e3:5.1f}us  b2b: search={ks:7.2f}us fit={kfit:6.2f}us  dev/pass={c['eval_ms'] / max(c['n_eval'], 1) * 1e3:7.2f}us  cand/q={cand:6.1f}",
          flush=True)
    kf.close()
    h.close()
