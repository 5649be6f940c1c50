#!/usr/bin/env python
"""Sensitivity of the graded outputs to what the oracle cannot pin (CPU only; the reference cannot be built here).

1. Eigen's fp32 summation order inside esti_plane's ColPivHouseholderQR (oracle_math.c "SUMMATION ORDER"): the full
   iterated update of BASELINE configs[1] (100k-point Avia scan vs 5M-point map) under each of the four orders.
   Reported against the default (SSE = Eigen 3.3.x on x86-64/SSE2, the reference's build): plane fits whose pabcd
   bits differ on the first pass, point_selected_surf flags that differ after the first and after the last pass,
   n_eff per pass, and the distance between the posteriors.
2. ikd-Tree's down-sampling box in float (its own arithmetic) vs the double voxel grid the oracle and the device
   use: points of the map that differ after one Add_Points of a scan's worth of points, at 0.5 (every launch file)
   and at marsim's 0.3.

    python tools/eigen_order_study.py [--small]      # writes a markdown table to stdout
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast_lio_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--small", action="store_true", help="200k-point map / 20k-point scan (seconds)")
ap.add_argument("--scans", type=int, default=4)
ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
args = ap.parse_args()
M, N = (200_000, 20_000) if args.small else (5_000_000, 100_000)
cfg = 1 if args.small else 2

scene = synth.make_scene(M, synth.CONFIG_SEED_BASE + cfg)
m = po.Map(scene.map_xyz)
rows = {o: dict(fits=0, fit_diff=0, flag1=0, flagN=0, neff=[], dpos=[], drot=[], dn=[]) for o in range(4)}
tot_pts = 0
for s in range(args.scans):
    pr = synth.make_problem(M, N, "avia", cfg=cfg, scan_seed=s, scene=scene)
    xp, P = synth.propagate_prior_cov(po.predict, pr.x_prior)
    res = {}
    for o in range(4):
        po.set_eigen_order(o)
        sc1 = po.Scan(pr.body, nthreads=args.threads)
        sc1.h_share_model(m, xp, True, False)
        sel1, nv1 = sc1.selected.copy(), sc1.normvec.copy()
        sc = po.Scan(pr.body, nthreads=args.threads)
        x, Pn, st = sc.update_iterated(m, xp, P)
        res[o] = (sel1, nv1, sc.selected.copy(), x, list(st.n_eff)[: st.passes])
    po.set_eigen_order(po.ORDER_SSE)
    ref = res[po.ORDER_SSE]
    tot_pts += N
    for o in range(4):
        sel1, nv1, selN, x, neff = res[o]
        both = (sel1 == 1) & (ref[0] == 1)
        r = rows[o]
        r["fits"] += int(both.sum())
        r["fit_diff"] += int((nv1[both].view(np.uint32)[:, :3] != ref[1][both].view(np.uint32)[:, :3]).any(axis=1).sum())
        r["flag1"] += int((sel1 != ref[0]).sum())
        r["flagN"] += int((selN != ref[2]).sum())
        r["neff"].append(neff)
        r["dpos"].append(float(np.linalg.norm(x[:3] - ref[3][:3])))
        r["drot"].append(float(2 * np.linalg.norm((x[3:7] * np.sign(x[6]) - ref[3][3:7] * np.sign(ref[3][6]))[:3])))
        r["dn"].append([a - b for a, b in zip(neff, ref[4])] if len(neff) == len(ref[4]) else ["schedule differs"])

print(f"### Eigen summation order: {args.scans} scans of {N} points vs the {M}-point map, full iterated update, vs ORDER_SSE\n")
print("| order | plane fits with different pabcd bits (pass 1) | point_selected_surf flips after pass 1 | flips after the last pass | "
      "n_eff difference per pass (scan 0) | max posterior position difference (m) | max rotation difference (rad) |")
print("|---|---|---|---|---|---|---|")
for o in range(4):
    r = rows[o]
    print(f"| {po.ORDER_NAMES[o]} | {r['fit_diff']} of {r['fits']} ({100.0 * r['fit_diff'] / max(r['fits'], 1):.1f} %) | "
          f"{r['flag1']} of {tot_pts} | {r['flagN']} of {tot_pts} | {r['dn'][0]} | {max(r['dpos']):.2e} | {max(r['drot']):.2e} |")

# ---- 2. the down-sampling box
print("\n### ikd-Tree down-sampling box: float corners (ikd-Tree's own) vs the double voxel grid (oracle, device)\n")
print("| downsample_size | map points | inserted | survivors (double grid) | survivors (float box) | points in one result but not the other |")
print("|---|---|---|---|---|---|")
pr = synth.make_problem(M, N, "avia", cfg=cfg, scan_seed=0, scene=scene)
sc = po.Scan(pr.body, nthreads=args.threads)
sc.h_share_model(m, pr.x_true, True, False)
rng = np.random.default_rng(1)
add = (sc.world + rng.normal(0, 0.05, sc.world.shape)).astype(np.float32)
base = scene.map_xyz[:: max(1, M // 400_000)].astype(np.float32)
for ds in (0.5, 0.3, 0.2):
    a = po.map_add(base, add, True, ds)
    b = po.map_add_floatbox(base, add, ds)
    sa = set(map(bytes, a.view(np.uint8).reshape(len(a), 12)))
    sb = set(map(bytes, b.view(np.uint8).reshape(len(b), 12)))
    print(f"| {ds} | {len(base)} | {len(add)} | {len(a)} | {len(b)} | {len(sa ^ sb)} |")
