"""Integer-logic model of k_search_tile (flh_search_tile.inc): bounding box, per-cell counts from the brick tables, prefix, flat copy,
per-query row ranges -- checked against brute force (every map point whose cell is within +-1 of the query's cell)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fast_lio_amd import synth, dist
from oracle import pyoracle as po

K_TILE_PTS, K_TILE_CELLS, STRIDE = 1152, 384, 80
pr = synth.make_problem(200000, 20000, 'avia', cfg=1)
mp = pr.map_xyz.astype(np.float32)
c = np.float32(1.5)
inv_c = np.float32(1.0) / c
lo = mp.min(axis=0) - np.float32(2) * c
org = lo.astype(np.float32)
def cell_of(p):
    f = (p - org) * inv_c
    fl = np.floor(f)
    return fl.astype(np.int64), (f - fl).astype(np.float32)
mc, _ = cell_of(mp)
dims = mc.max(axis=0) + 3
nx, ny, nz = map(int, dims)
bkey = ((mc[:, 2] >> 2) << 20) | ((mc[:, 1] >> 2) << 10) | (mc[:, 0] >> 2)
lcl = ((mc[:, 2] & 3) << 4) | ((mc[:, 1] & 3) << 2) | (mc[:, 0] & 3)
key = (bkey << 6) | lcl
order = np.argsort(key, kind='stable')
ks = key[order]
pts_sorted = mp[order]
ub, first = np.unique(ks >> 6, return_index=True)
rank_of = {int(k): r for r, k in enumerate(ub)}
first = np.append(first, len(ks))
starts = np.zeros((len(ub), STRIDE), np.int64)
for r in range(len(ub)):
    seg = ks[first[r]:first[r + 1]] & 63
    for l in range(65):
        starts[r, l] = first[r] + np.searchsorted(seg, l, side='left')
print('bricks', len(ub), 'grid', nx, ny, nz)

body = pr.body[dist.morton_order(pr.body)]
w = po.points_body_to_world(pr.x_prior, body).astype(np.float32)
qc, qf = cell_of(w)
cellkey_map = (mc[:, 2] * ny + mc[:, 1]) * nx + mc[:, 0]
from collections import defaultdict
cell_pts = defaultdict(list)
for i, ck in enumerate(cellkey_map[order]):
    cell_pts[int(ck)].append(i)   # positions in sorted storage

def brute(q):
    cx, cy, cz = qc[q]
    out = []
    for z in range(cz - 1, cz + 2):
        for y in range(cy - 1, cy + 2):
            for x in range(cx - 1, cx + 2):
                if 0 <= x < nx and 0 <= y < ny and 0 <= z < nz:
                    out += cell_pts.get((z * ny + y) * nx + x, [])
    return sorted(out)

nfit = nblk = 0
for blk in range(0, len(body) // 64, 7):
    qs = list(range(blk * 64, blk * 64 + 64))
    nblk += 1
    nxa = np.maximum(qc[qs, 0] - 1, 0); nxb = np.minimum(qc[qs, 0] + 1, nx - 1)
    nya = np.maximum(qc[qs, 1] - 1, 0); nyb = np.minimum(qc[qs, 1] + 1, ny - 1)
    nza = np.maximum(qc[qs, 2] - 1, 0); nzb = np.minimum(qc[qs, 2] + 1, nz - 1)
    has = (nxa <= nxb) & (nya <= nyb) & (nza <= nzb)
    if not has.any():
        continue
    x0, y0, z0 = nxa[has].min(), nya[has].min(), nza[has].min()
    Wx, Wy, Wz = nxb[has].max() - x0 + 1, nyb[has].max() - y0 + 1, nzb[has].max() - z0 + 1
    ncell = Wx * Wy * Wz
    if ncell > K_TILE_CELLS:
        continue
    cstart = np.zeros(ncell + 1, np.int64); cgpos = np.zeros(ncell, np.int64)
    bx0 = x0 >> 2
    nbx = ((x0 + Wx - 1) >> 2) - bx0 + 1
    for it in range(Wy * Wz * nbx):
        sx, row = it % nbx, it // nbx
        ty, tz = row % Wy, row // Wy
        bx = bx0 + sx
        xa, xb = max(x0, bx << 2), min(x0 + Wx - 1, (bx << 2) + 3)
        y, z = y0 + ty, z0 + tz
        k = ((z >> 2) << 20) | ((y >> 2) << 10) | (xa >> 2)
        r = rank_of.get(int(k))
        if r is None:
            continue
        l0 = ((z & 3) << 4) | ((y & 3) << 2) | (xa & 3)
        e = [starts[r, l0 + min(kk, xb - xa + 1)] for kk in range(5)]
        t0 = (tz * Wy + ty) * Wx + (xa - x0)
        for kk in range(4):
            if kk <= xb - xa:
                cstart[t0 + kk] = e[kk + 1] - e[kk]
                cgpos[t0 + kk] = e[kk]
    cnt = cstart[:ncell].copy()
    cstart[:] = np.concatenate([[0], np.cumsum(cnt)])
    P = cstart[ncell]
    if P > K_TILE_PTS:
        continue
    nfit += 1
    tile = np.zeros(P, np.int64)
    for p in range(P):
        lo_, hi_ = 0, ncell
        while hi_ - lo_ > 1:
            mid = (lo_ + hi_) >> 1
            if cstart[mid] <= p: lo_ = mid
            else: hi_ = mid
        tile[p] = cgpos[lo_] + (p - cstart[lo_])
    for j, q in enumerate(qs):
        got = []
        cx, cy, cz = qc[q]
        for r in range(9):
            dy, dz = r % 3 - 1, r // 3 - 1
            y, z = cy + dy, cz + dz
            if not (has[j] and 0 <= y < ny and 0 <= z < nz):
                continue
            t0 = ((z - z0) * Wy + (y - y0)) * Wx + (nxa[j] - x0)
            ps, pe = cstart[t0], cstart[t0 + (nxb[j] - nxa[j]) + 1]
            got += list(tile[ps:pe])
        assert sorted(got) == brute(q), (blk, q)
print('blocks checked', nblk, 'fit', nfit, 'all candidate sets equal brute force')
