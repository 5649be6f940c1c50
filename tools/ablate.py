#!/usr/bin/env python
"""Developer tool: back-to-back timing of search-kernel ablations (results are NOT valid searches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast_lio_amd import capi, synth
from bench import CONFIGS
M, N, sensor = CONFIGS[2]
scene = synth.make_scene(M, synth.CONFIG_SEED_BASE + 2)
pr = synth.make_problem(M, N, sensor, cfg=2, scan_seed=0, scene=scene)
xp, P = synth.propagate_prior_cov(capi.predict_fn, pr.x_prior)
for lpq in (1204, 2204, 204):
    h = capi.Handle(cell_size=1.0, lanes_per_query=lpq)
    h.map_build(scene.map_xyz)
    h.scan_upload(pr.body)
    for _ in range(3):
        ms = h.time_kernel(0, xp, False, 20)
    print(f"variant {lpq}: {ms*1e3:8.2f} us per launch (b2b)", flush=True)
    h.close()
