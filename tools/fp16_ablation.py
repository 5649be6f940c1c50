#!/usr/bin/env python
"""BASELINE configs[4]: fp32 vs fp16 plane fit (flh_config.plane_fit_dtype) -- flag-mismatch rate against the bit-exact fp32
path, posterior difference of the full update, and the fit kernel's time.  Needs a GPU.

    python tools/fp16_ablation.py [--config 5]         # markdown table on stdout
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from fast_lio_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=5)
ap.add_argument("--scans", type=int, default=3)
args = ap.parse_args()
M, N, sensor = CONFIGS[args.config]
scene = synth.make_scene(M, synth.CONFIG_SEED_BASE + args.config)
hs = {d: capi.Handle(plane_fit_dtype=d) for d in (0, 1)}
for h in hs.values():
    h.map_build(scene.map_xyz)
rows = []
for s in range(args.scans):
    pr = synth.make_problem(M, N, sensor, cfg=args.config, scan_seed=s, scene=scene)
    xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
    out = {}
    for d, h in hs.items():
        h.scan_upload(pr.body)
        h.eval(xp, True, False)
        sel1 = h.fetch_selected().astype(bool)
        kf = capi.Esekf(h, max_iter=3)
        kf.change_x(xp)
        kf.change_P(P)
        st = kf.update(0.001)
        fit_us = h.time_kernel(1, xp, False, 50) * 1e3
        out[d] = (sel1, h.fetch_selected().astype(bool), kf.get_x(), list(st.n_eff)[: st.passes], fit_us)
        kf.close()
    a, b = out[0], out[1]
    rows.append((s, N, (a[0] != b[0]).mean(), (a[1] != b[1]).mean(), a[3], b[3], np.linalg.norm(a[2][:3] - b[2][:3]),
                 np.linalg.norm(b[2][:3] - pr.x_true[:3]) - np.linalg.norm(a[2][:3] - pr.x_true[:3]), a[4], b[4]))
print(f"### fp32 vs fp16 plane fit, BASELINE configs[{args.config - 1}]: {sensor} {N}-pt scans vs {M}-pt map\n")
print("| scan | flag mismatch after pass 1 | after the last pass | n_eff per pass fp32 | n_eff per pass fp16 | posterior position difference (m) | "
      "change of the position error vs truth (m) | k_fit fp32 (us) | k_fit fp16 (us) |")
print("|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r[0]} | {100 * r[2]:.2f} % | {100 * r[3]:.2f} % | {r[4]} | {r[5]} | {r[6]:.2e} | {r[7]:+.2e} | {r[8]:.2f} | {r[9]:.2f} |")
