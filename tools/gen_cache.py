"""Pre-generates bench.py's scan cache for one BASELINE config (the 20M / 50M-point scenes take tens of seconds per scan):
   python tools/gen_cache.py CONFIG COUNT [SEED_BASE]
writes .bench_cache/scans_cfg{CONFIG}_n{COUNT}_b{SEED_BASE}.npz exactly as bench.py's gen() would (same seeds, same bytes).
The directory is git-ignored but travels to the GPU box with the repository snapshot."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fast_lio_amd import capi, synth  # noqa: E402

CONFIGS = {1: (200_000, 20_000, "avia"), 2: (5_000_000, 100_000, "avia"), 3: (10_000_000, 60_000, "velodyne"),
           4: (20_000_000, 130_000, "ouster64"), 5: (50_000_000, 200_000, "mid360")}
cfg, count = int(sys.argv[1]), int(sys.argv[2])
base = int(sys.argv[3]) if len(sys.argv) > 3 else 0
M, N, sensor = CONFIGS[cfg]
scene = synth.make_scene(M, synth.CONFIG_SEED_BASE + cfg)
with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
    full = list(ex.map(lambda s_: synth.make_problem(M, N, sensor, cfg=cfg, scan_seed=base + s_, scene=scene), range(count)))
pri = [synth.propagate_prior_cov(capi.predict_fn, p.x_prior) for p in full]
d = os.path.join(ROOT, ".bench_cache")
os.makedirs(d, exist_ok=True)
out = os.path.join(d, f"scans_cfg{cfg}_n{count}_b{base}.npz")
np.savez(out + ".tmp.npz", body=np.stack([p.body for p in full]), x_prior=np.stack([p.x_prior for p in full]),
         x=np.stack([np.ascontiguousarray(x, np.float64) for x, _ in pri]), P=np.stack([np.ascontiguousarray(P, np.float64) for _, P in pri]))
os.replace(out + ".tmp.npz", out)
print(out, os.path.getsize(out) >> 20, "MiB")
