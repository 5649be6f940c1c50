#!/usr/bin/env python
"""Developer tool: per-phase wave cycle counts of the search kernels (-DFLH_PHASES build in a separate library).

    python tools/phases.py --build-only      # here (cross-compile)
    python tools/phases.py [--config 2]      # on the GPU box
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast_lio_amd import _build  # noqa: E402

_build.LIB = os.path.join(_build.LIBDIR, "libfastlio_hip_phases.so")
_build.FLAGS = _build.FLAGS + ["-DFLH_PHASES"]

ap = argparse.ArgumentParser()
ap.add_argument("--build-only", action="store_true")
ap.add_argument("--M", type=int, default=5000000)
ap.add_argument("--N", type=int, default=100000)
ap.add_argument("--cell", type=float, default=1.5)
args = ap.parse_args()
if args.build_only:
    print(_build.build(force=True))
    sys.exit(0)

from fast_lio_amd import capi, synth  # noqa: E402

pr = synth.make_problem(args.M, args.N, "avia", cfg=2)
h = capi.Handle(cell_size=args.cell)
h.map_build(pr.map_xyz)
h.scan_upload(pr.body)
xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
h.eval(xp, True, False)
h.enable_stats(True)
for name, x in (("prior", xp), ("truth", pr.x_true)):
    print(f"--- search at the {name} state", file=sys.stderr)
    for _ in range(2):
        h.eval(x, True, False)
    print("--- no-search pass", file=sys.stderr)
    h.eval(x, False, False)
