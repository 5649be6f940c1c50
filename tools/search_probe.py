#!/usr/bin/env python
"""Developer tool: one search-stage variant under rocprofv3 (kernel trace or --pmc): a few search passes at the prior
state (first pass of a scan) and at the true state (converged pass), nothing else.  Needs a GPU."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast_lio_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=5_000_000)
ap.add_argument("--N", type=int, default=100_000)
ap.add_argument("--sensor", default="avia")
ap.add_argument("--cfg", type=int, default=2)
ap.add_argument("--lpq", type=int, default=4)
ap.add_argument("--pass-kernel", type=int, default=-1)
ap.add_argument("--cell", type=float, default=1.5)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--state", default="both", choices=["prior", "truth", "both"])
args = ap.parse_args()

pr = synth.make_problem(args.M, args.N, args.sensor, cfg=args.cfg)
xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
h = capi.Handle(cell_size=args.cell, lanes_per_query=args.lpq, pass_kernel=args.pass_kernel)
h.map_build(pr.map_xyz)
h.scan_upload(pr.body)
h.set_timing_stride(0)
for name, x in (("prior", xp), ("truth", pr.x_true)):
    if args.state not in (name, "both"):
        continue
    for _ in range(args.reps):
        h.eval(x, True, False)
    for _ in range(args.reps):
        h.eval(x, False, False)
h.close()
