#!/usr/bin/env python
"""Developer tool: build the library with extra -D flags under another name and run search passes with it (wrap in rocprofv3 for
kernel times): A/B experiments that never touch the product library.

    python tools/variant.py --name capT --define FLH_EXP_CAP_T=96 --build-only     # here (cross-compile)
    rocprofv3 --kernel-trace --stats ... -- python tools/variant.py --name capT    # on the GPU box
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast_lio_amd import _build  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--name", default="")
ap.add_argument("--define", action="append", default=[])
ap.add_argument("--build-only", action="store_true")
ap.add_argument("--M", type=int, default=5_000_000)
ap.add_argument("--N", type=int, default=100_000)
ap.add_argument("--cfg", type=int, default=2)
ap.add_argument("--sensor", default="avia")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--pass-kernel", type=int, default=-1)
ap.add_argument("--lpq", type=int, default=4)
args = ap.parse_args()
if args.name:
    _build.LIB = os.path.join(_build.LIBDIR, f"libfastlio_hip_{args.name}.so")
    _build.FLAGS = _build.FLAGS + ["-D" + d for d in args.define]
if args.build_only:
    print(_build.build(force=True))
    sys.exit(0)
if args.name and not os.path.exists(_build.LIB):
    sys.exit(f"{_build.LIB} missing: build it first (--build-only)")
_build.needs_build = lambda: False

from fast_lio_amd import capi, synth  # noqa: E402

pr = synth.make_problem(args.M, args.N, args.sensor, cfg=args.cfg)
xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
h = capi.Handle(lanes_per_query=args.lpq, pass_kernel=args.pass_kernel)
h.map_build(pr.map_xyz)
h.scan_upload(pr.body)
h.set_timing_stride(0)
for x in (xp, pr.x_true):
    for _ in range(args.reps):
        h.eval(x, True, False)
        h.eval(x, False, False)
sel = h.selected() if hasattr(h, "selected") else None
print("done", "" if sel is None else int(sel.sum()))
h.close()
