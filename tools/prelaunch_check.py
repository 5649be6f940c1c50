#!/usr/bin/env python
"""GPU box, developer build FLH_EXP_PRELAUNCH (the no-search passes of an update are enqueued ahead of their states and fed
through a mailbox, fast_lio_amd/csrc/exp/): is the result the same, bit for bit, and what does it buy?

    python tools/variant.py --name prelaunch --define FLH_EXP_PRELAUNCH --build-only          # here (cross-compile)
    FLH_LIB=fast_lio_amd/lib/libfastlio_hip_prelaunch.so python tools/prelaunch_check.py      # on the GPU box

0. kernel    one no-search evaluation through the mailbox against the same evaluation launched the usual way: same bits.
1. parity   the same scans updated with the switch off (plain launches) and on: posterior state and covariance, pass schedule,
            n_eff of every pass, point_selected_surf must be IDENTICAL (the mailbox kernel runs k_fit<1,false,2>'s statements on
            the same units in the same tree); the counters must show that the mailbox was really used.
2. timing   flh_esekf_run_scans over the same scans, alternating off / on, three times each: scans/s, ms per no-search pass.
3. lateness the host sleeps 30 ms between arming and posting (a stand-in for a descheduled thread): the waiting kernel gives up,
            the pass is launched the usual way, same bits.
Exit code 0 = all comparisons equal."""
import argparse
import ctypes as C
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
args = ap.parse_args()

L = capi.lib()
if not hasattr(L, "flh_exp_prelaunch_switch"):
    sys.exit("this library was not built with -DFLH_EXP_PRELAUNCH (set FLH_LIB to the variant)")
L.flh_exp_prelaunch_switch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
L.flh_exp_prelaunch.argtypes = [C.c_void_p, C.c_int]


def switch(h, on):
    out = (C.c_uint64 * 4)()
    if L.flh_exp_prelaunch_switch(h.ptr, on, out) != 0:
        raise RuntimeError(L.flh_last_error().decode())
    return dict(zip(("armed", "go", "abort", "gone"), [int(v) for v in out]))


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
bad = 0


def update_all(on):
    switch(h, 1 if on else 0)
    res = []
    for k, pr in enumerate(probs):
        h.scan_upload(pr.body)
        kf.change_x(priors[k][0])
        kf.change_P(priors[k][1])
        st = kf.update()
        res.append((kf.get_x().copy(), kf.get_P().copy(), st.passes, st.searches, [st.n_eff[i] for i in range(st.passes)],
                    [st.pass_search[i] for i in range(st.passes)], h.fetch_selected().copy()))
    return res


# ---- 0. one evaluation, kernel against kernel: k_fit<1,false,2> and k_fit_mb at the same state after the same search
switch(h, 1)
h.scan_upload(probs[0].body)
xa, xb = priors[0][0], probs[0].x_true
ref_s = h.eval(xa, True, False)
ref_n = h.eval(xb, False, False)
ca = switch(h, -1)
L.flh_exp_prelaunch(h.ptr, 1)
got_s = h.eval(xa, True, False)     # arms the next pass's kernel
got_n = h.eval(xb, False, False)    # handed over through the mailbox
L.flh_exp_prelaunch(h.ptr, 0)
cb = switch(h, -1)
one_ok = (np.array_equal(got_s[0], ref_s[0]) and np.array_equal(got_n[0], ref_n[0]) and np.array_equal(got_n[1], ref_n[1])
          and got_n[2] == ref_n[2] and got_n[3] == ref_n[3])
print("one evaluation: go", cb["go"] - ca["go"], "abort", cb["abort"] - ca["abort"], "->", "identical" if one_ok else
      f"DIFFERENT (max |dHTH| {np.abs(got_n[0] - ref_n[0]).max():.3e} of {np.abs(ref_n[0]).max():.3e}, n_eff {got_n[2]} vs {ref_n[2]})")
if not one_ok or cb["go"] == ca["go"]:
    bad += 1

# ---- 1. parity
c0 = switch(h, -1)
ref = update_all(False)
c1 = switch(h, -1)
got = update_all(True)
c2 = switch(h, -1)
print("counters  off:", {k: c1[k] - c0[k] for k in c1}, " on:", {k: c2[k] - c1[k] for k in c2})
if c1["go"] != c0["go"]:
    print("FAIL: the mailbox was used with the switch off"); bad += 1
if c2["go"] == c1["go"]:
    print("FAIL: the mailbox was never used with the switch on"); bad += 1
for k, (a, b) in enumerate(zip(ref, got)):
    same = (a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes() and a[2:6] == b[2:6] and np.array_equal(a[6], b[6]))
    print(f"scan {k}: passes {b[2]} searches {b[3]} schedule {b[5]} n_eff {b[4]}  {'identical' if same else 'DIFFERENT'}")
    if not same:
        bad += 1
        print("   max |dx|", np.abs(a[0] - b[0]).max(), " max |dP|", np.abs(a[1] - b[1]).max(), " flags differing", int((a[6] != b[6]).sum()))

# ---- 3. lateness: arm by hand (the bracket open), wait past the forwarder's patience, then evaluate
switch(h, 1)
h.scan_upload(probs[0].body)
x = priors[0][0]
o_ref_s = h.eval(x, True, False)
o_ref_n = h.eval(x, False, False)
L.flh_exp_prelaunch(h.ptr, 1)
o_s = h.eval(x, True, False)        # arms the next pass's kernel
time.sleep(0.03)                    # ... which gives up after 20 ms
o_n = h.eval(x, False, False)       # mail posted to nobody: collect_granules notices and launches the usual way
L.flh_exp_prelaunch(h.ptr, 0)
c3 = switch(h, -1)
late_ok = np.array_equal(o_n[0], o_ref_n[0]) and np.array_equal(o_n[1], o_ref_n[1]) and o_n[2] == o_ref_n[2] and np.array_equal(o_s[0], o_ref_s[0])
print("lateness: gone", c3["gone"] - c2["gone"], "->", "identical" if late_ok else "DIFFERENT")
if not late_ok or c3["gone"] == c2["gone"]:
    bad += 1

# ---- 2. timing
bodies = [np.ascontiguousarray(pr.body, np.float32) for pr in probs]
jobs = capi.Esekf.make_jobs(bodies, priors)
for rep in range(3):
    for on in (0, 1):
        switch(h, on)
        kf.run_scans(jobs, 0, 10)
        t0 = time.perf_counter()
        rs = kf.run_scans(jobs, 0, args.steps)
        dt = time.perf_counter() - t0
        print(f"rep {rep} prelaunch {'on ' if on else 'off'}: {args.steps / dt:8.1f} scans/s  {1e3 * dt / args.steps:.4f} ms/scan  "
              f"searching pass {1e3 * rs.ms_search_passes / max(rs.n_search_passes, 1):.1f} us  "
              f"no-search pass {1e3 * rs.ms_nosearch_passes / max(rs.n_nosearch_passes, 1):.1f} us  passes/scan {rs.passes / rs.scans:.2f}")
print("counters at the end:", switch(h, -1))
kf.close()
h.close()
print("RESULT:", "all identical" if bad == 0 else f"{bad} FAILED")
sys.exit(0 if bad == 0 else 1)
