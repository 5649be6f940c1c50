"""How large would a shared LDS tile be?  For groups of G consecutive (Morton-ordered) queries of a synthetic BASELINE scan: cells and
map points inside the bounding box of their 3x3x3 neighbourhoods, the share of groups that fit a given LDS budget, and how many times
each tile point would be used (sum of the per-query candidate counts / tile points).  CPU only.   python tools/tile_stats.py [config]"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fast_lio_amd import synth, dist
from oracle import pyoracle as po

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
M, N, sensor = {1: (200_000, 20_000, 'avia'), 2: (5_000_000, 100_000, 'avia'), 3: (10_000_000, 60_000, 'velodyne')}[cfg]
t0 = time.time()
pr = synth.make_problem(M, N, sensor, cfg=cfg)
print('gen', time.time() - t0)
body = pr.body
perm = dist.morton_order(body)
bs = body[perm]
w = po.points_body_to_world(pr.x_prior, bs).astype(np.float64)
mp = pr.map_xyz.astype(np.float64)
c = 1.5
org = mp.min(axis=0) - 2 * c
mc = np.floor((mp - org) / c).astype(np.int64)
dims = mc.max(axis=0) + 3
key = (mc[:, 2] * dims[1] + mc[:, 1]) * dims[0] + mc[:, 0]
cnt = np.bincount(key, minlength=int(dims.prod()))
print('map cells occupied', (cnt > 0).sum(), 'mean pts per occupied cell', cnt[cnt > 0].mean())
qc = np.floor((w - org) / c).astype(np.int64)
qc = np.clip(qc, 1, dims - 2)
# 3D prefix (integral image) of counts for fast box sums
vol = cnt.reshape(dims[2], dims[1], dims[0]).astype(np.int64)
I = np.zeros((dims[2] + 1, dims[1] + 1, dims[0] + 1), np.int64)
I[1:, 1:, 1:] = vol.cumsum(0).cumsum(1).cumsum(2)
def boxsum(lo, hi):  # inclusive cell boxes, arrays (n,3) in x,y,z
    x0, y0, z0 = lo[:, 0], lo[:, 1], lo[:, 2]
    x1, y1, z1 = hi[:, 0] + 1, hi[:, 1] + 1, hi[:, 2] + 1
    return (I[z1, y1, x1] - I[z0, y1, x1] - I[z1, y0, x1] - I[z1, y1, x0] + I[z0, y0, x1] + I[z0, y1, x0] + I[z1, y0, x0] - I[z0, y0, x0])
per_query = boxsum(qc - 1, qc + 1)
print('candidates per query (3x3x3): mean %.1f p50 %d p90 %d p99 %d' % (per_query.mean(), *np.percentile(per_query, [50, 90, 99])))
for G in (16, 32, 64):
    n = (N // G) * G
    q = qc[:n].reshape(-1, G, 3)
    lo = q.min(axis=1) - 1
    hi = q.max(axis=1) + 1
    ext = hi - lo + 1
    cells = ext.prod(axis=1)
    pts = boxsum(lo, hi)
    need = per_query[:n].reshape(-1, G).sum(axis=1)
    print(f'G={G}: tile cells p50 {np.percentile(cells,50):.0f} p90 {np.percentile(cells,90):.0f} p99 {np.percentile(cells,99):.0f} max {cells.max()};'
          f' tile pts p50 {np.percentile(pts,50):.0f} p90 {np.percentile(pts,90):.0f} p99 {np.percentile(pts,99):.0f} max {pts.max()};'
          f' sum of per-query candidates / tile pts: mean {np.mean(need/np.maximum(pts,1)):.1f}')
    for capc, capp in ((256, 1024), (512, 1536), (1024, 2048), (2048, 4096)):
        ok = (cells <= capc) & (pts <= capp)
        print(f'    cap cells {capc} pts {capp}: {100*ok.mean():.1f}% of groups fit')
