"""Integer-logic model of k_search_wtile (flh_search_wtile.inc): a wave's bounding box, per-cell counts from the brick tables, prefix,
the per-(row, brick segment) copy, per-query row ranges -- checked against brute force (every map point whose cell is within +-1
of the query's cell); prints the share of waves that fit the tile for 2 and 4 lanes per query.
  python tools/tile_model.py [config] [stride]      config 1 (default, 20k-pt scan) or 2 (100k-pt scan vs 5M-pt map)"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fast_lio_amd import synth, dist
from oracle import pyoracle as po

STRIDE = 80
CFG = int(sys.argv[1]) if len(sys.argv) > 1 else 1
STEP = int(sys.argv[2]) if len(sys.argv) > 2 else 7
pr = synth.make_problem(200000, 20000, 'avia', cfg=1) if CFG == 1 else synth.make_problem(5000000, 100000, 'avia', cfg=2)
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

def check(Q, TP, TC, IT):
    nfit = nblk = 0
    for blk in range(0, len(body) // Q, STEP):
        qs = list(range(blk * Q, blk * Q + Q))
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
        bx0 = x0 >> 2
        nbx = ((x0 + Wx - 1) >> 2) - bx0 + 1
        items = Wy * Wz * nbx
        if ncell > TC or items > IT:
            continue
        cstart = np.zeros(TC + 8, np.int64)
        item = []
        inv_nbx, inv_wy = np.float32(1) / np.float32(max(nbx, 1)), np.float32(1) / np.float32(max(Wy, 1))
        for it in range(items):
            row = int((np.float32(it) + np.float32(0.5)) * inv_nbx); sx = it - row * nbx
            tz = int((np.float32(row) + np.float32(0.5)) * inv_wy); ty = row - tz * Wy
            assert (sx, row, ty, tz) == (it % nbx, it // nbx, row % Wy, row // Wy)
            bx = bx0 + sx
            xa, xb = max(x0, bx << 2), min(x0 + Wx - 1, (bx << 2) + 3)
            y, z = y0 + ty, z0 + tz
            t0 = (tz * Wy + ty) * Wx + (xa - x0)
            k = ((z >> 2) << 20) | ((y >> 2) << 10) | (xa >> 2)
            r = rank_of.get(int(k))
            gpos = npts = 0
            if r is not None:
                l0 = ((z & 3) << 4) | ((y & 3) << 2) | (xa & 3)
                e = [starts[r, l0 + min(kk, xb - xa + 1)] for kk in range(5)]
                for kk in range(4):
                    if kk <= xb - xa:
                        cstart[t0 + kk] = min(e[kk + 1] - e[kk], TP + 1)
                gpos, npts = e[0], e[xb - xa + 1] - e[0]
            item.append((gpos, t0, min(npts, 0xFFFF)))
        cnt = cstart[:TC].copy()
        P = int(cnt.sum())
        if P > TP:
            continue
        cstart[:TC] = np.concatenate([[0], np.cumsum(cnt)])[:TC]
        cstart[TC] = P
        nfit += 1
        # segments in tile order, their first tile positions as a bitmap, the copy by popcount (flh_search_wtile.inc)
        seg_tab = [(gpos, int(cstart[t0])) for gpos, t0, n in item if n > 0]
        assert all(seg_tab[i][1] < seg_tab[i + 1][1] for i in range(len(seg_tab) - 1))
        PPL = TP // 64
        bits = [0] * PPL
        for gpos, toff in seg_tab:
            bits[toff >> 6] |= 1 << (toff & 63)
        before = [0] * PPL
        for k in range(1, PPL):
            before[k] = before[k - 1] + bin(bits[k - 1]).count("1")
        tile = -np.ones(P, np.int64)
        for lane in range(64):
            for k in range(PPL):
                pp = lane + 64 * k
                if pp < P:
                    si = before[k] + bin(bits[k] & ((1 << (lane + 1)) - 1)).count("1") - 1
                    gpos, toff = seg_tab[si]
                    tile[pp] = gpos + (pp - toff)
        ref = -np.ones(P, np.int64)
        for gpos, t0, n in item:
            toff = cstart[t0]
            for j in range(n):
                ref[toff + j] = gpos + j
        assert (tile == ref).all() and (tile >= 0).all()
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
    print(f'Q={Q} (TP={TP}, TC={TC}, IT={IT}): waves checked {nblk}, fit {nfit} ({100.0 * nfit / max(nblk, 1):.1f} %), all candidate sets equal brute force')


check(32, 1024, 512, 192)
check(16, 640, 256, 128)
