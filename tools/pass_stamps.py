#!/usr/bin/env python
"""Developer tool (GPU box): where the time of the one-launch pass goes.  Needs the library built with -DFLH_PASS_STAMPS
(python tools/variant.py --name stamps --define FLH_PASS_STAMPS --build-only; FLH_LIB=.../libfastlio_hip_stamps.so).
Prints, for a search at the prior and one at the true state of BASELINE configs[1], the distribution over waves / workgroups of
the phase end times of k_pass relative to the kernel's first wave start (microseconds)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast_lio_amd import capi, synth  # noqa: E402

M, N = 5_000_000, 100_000
pr = synth.make_problem(M, N, "avia", cfg=2)
xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
h = capi.Handle()
h.map_build(pr.map_xyz)
h.scan_upload(pr.body)
h.set_timing_stride(0)
nw = (N + 63) // 64 * 4


def pc(v, name):
    if len(v) == 0:
        print(f"  {name:34s} (none)")
        return
    q = np.percentile(v, [0, 25, 50, 75, 95, 99, 100])
    print(f"  {name:34s} n={len(v):5d}  p0 {q[0]:6.2f}  p25 {q[1]:6.2f}  p50 {q[2]:6.2f}  p75 {q[3]:6.2f}  p95 {q[4]:6.2f}  p99 {q[5]:6.2f}  max {q[6]:6.2f}")


for name, x in (("prior (first search of a scan)", xp), ("true state (a later search)", pr.x_true)):
    for rep in range(3):
        h.eval(x, True, False)
    ok, st = h.debug_pass_stamps(nw)
    if not ok:
        sys.exit("library not built with -DFLH_PASS_STAMPS")
    t = st.astype(np.float64) / 100.0  # us
    t0 = t[:, 0][t[:, 0] > 0].min()
    rel = np.where(t > 0, t - t0, np.nan)
    fit = ~np.isnan(rel[:, 4]) & (st[:, 4] > st[:, 0])  # the fit waves (stamps of this launch: later than the wave's own start)
    red = fit & (st[:, 7] > st[:, 6])
    print(name)
    pc(rel[:, 0], "wave start")
    pc(rel[:, 1], "wave: phase A done")
    pc(rel[:, 1] - rel[:, 0], "wave: phase A duration")
    pc(rel[fit, 2], "workgroup: phase A done (barrier)")
    pc(rel[fit, 3] - rel[fit, 2], "workgroup: phase B duration")
    pc(rel[fit, 4] - rel[fit, 3], "fit wave: fit + Gram duration")
    pc(rel[fit, 5] - rel[fit, 4], "fit wave: partial store + drain")
    pc(rel[fit, 6] - rel[fit, 5], "fit wave: ticket round trip")
    pc(rel[fit, 6], "fit wave: done (ticket taken)")
    pc(rel[red, 7] - rel[red, 6], "reducer: group sum + publish")
    pc(rel[red, 7], "reducer: published")
    last = np.nanmax(rel[:, :8])
    print(f"  last stamp of the launch: {last:.2f} us after the first wave's start")
    # which waves are the slow ones?  by length of the longest candidate list (trips of 32 candidates per query), by place
    durA = rel[:, 1] - rel[:, 0]
    tmax, nopen = st[:, 10].astype(np.int64), st[:, 11].astype(np.int64)
    hw, xcc, unit = st[:, 8].astype(np.int64), st[:, 9].astype(np.int64) & 0xF, (st[:, 9] >> np.uint64(32)).astype(np.int64)
    if not unit.any():  # a build that does not stamp the unit: the product's mapping, the END of the Morton order first
        unit = (nw // 4 - 1) - np.arange(len(unit)) // 4
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    trips = (tmax + 31) // 32
    print("  phase A duration by trips of the wave's longest query:")
    for k in range(0, int(trips.max()) + 1):
        m = trips == k
        if m.any():
            print(f"    trips {k}: n={int(m.sum()):5d}  mean {np.nanmean(durA[m]):6.2f}  p95 {np.nanpercentile(durA[m], 95):6.2f}  max {np.nanmax(durA[m]):6.2f}")
    print("  phase A duration by XCC:", "  ".join(f"{x}: {np.nanmean(durA[xcc == x]):.2f}/{np.nanmax(durA[xcc == x]):.2f}" for x in sorted(set(xcc.tolist()))))
    order = np.argsort(-np.nan_to_num(durA))[:16]
    print("  slowest waves (phase A): wave block unit dur start tmax open xcc se sh cu simd")
    for w in order:
        print(f"    {w:5d} {w // 4:5d} {unit[w]:5d} {durA[w]:6.2f} {rel[w, 0]:5.2f} {tmax[w]:4d} {nopen[w]:3d}  {xcc[w]} {se[w]} {sh[w]} {cu[w]:2d} {simd[w]}")
    # by dispatch position (blockIdx) and by unit (position in the scan's Morton order): where the slow workgroups are, and when they start
    nb = (len(durA) + 3) // 4
    wgA = np.nanmax(rel[: nb * 4, 2].reshape(nb, 4), axis=1)          # the workgroup's phase A done
    wgF = np.nanmax(np.nan_to_num(rel[: nb * 4, 6], nan=-1.0).reshape(nb, 4), axis=1)  # its fit wave done
    wgS = np.nanmin(rel[: nb * 4, 0].reshape(nb, 4), axis=1)
    wgU = unit[: nb * 4].reshape(nb, 4)[:, 0]
    print("  deciles of the DISPATCH order (blockIdx): start / phase A done / fit done (means), max fit done")
    for d in range(10):
        m = slice(d * nb // 10, (d + 1) * nb // 10)
        print(f"    {d}: {np.nanmean(wgS[m]):5.2f} / {np.nanmean(wgA[m]):6.2f} / {np.nanmean(wgF[m]):6.2f}   max {np.nanmax(wgF[m]):6.2f}")
    print("  deciles of the UNIT index (Morton order of the scan): phase A duration of the workgroup (mean / max), mean dispatch position")
    o = np.argsort(wgU)
    for d in range(10):
        m = o[d * nb // 10:(d + 1) * nb // 10]
        print(f"    {d}: {np.nanmean(wgA[m] - wgS[m]):6.2f} / {np.nanmax(wgA[m] - wgS[m]):6.2f}   {np.mean(m):7.1f}")
    # waves per (xcc, se, sh, cu): is a CU that got more workgroups slower?
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    cnt = np.bincount(cuid)
    per = cnt[cuid]
    for k in sorted(set(per.tolist())):
        m = per == k
        print(f"    waves on CUs holding {k:2d} waves: n={int(m.sum()):5d}  mean phase A {np.nanmean(durA[m]):6.2f}  max {np.nanmax(durA[m]):6.2f}")
h.close()
