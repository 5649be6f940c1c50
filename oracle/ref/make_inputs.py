#!/usr/bin/env python
"""Seeded inputs for the oracle/_ref binaries + the oracle's own outputs for the same inputs (oracle/_ref/*.bin).

planes_in.bin   : 6000 five-point sets (random planes at 1..450 m, co-planar to a few cm of noise; one sixth with one
                  point pushed to 0.1 m +- 2 mm off the plane, where esti_plane's verdict is decided by the last bits)
planes_orc.npz  : pabcd bits + verdict of the oracle under each of the four summation orders
iekf_in.bin     : recorded full updates (prior, P, per-pass rows) of small synthetic problems, with and without the
                  gain-form branch (n_eff < 23)
iekf_orc.npz    : the oracle's posteriors
"""
import ctypes as C
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from fast_lio_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "oracle", "_ref")
os.makedirs(OUT, exist_ok=True)


def plane_sets(n=6000, seed=20240807):
    rng = np.random.default_rng(seed)
    out = np.zeros((n, 5, 3), np.float32)
    for i in range(n):
        c = rng.uniform(-1, 1, 3)
        c = c / np.linalg.norm(c) * rng.uniform(1.0, 450.0)
        nrm = rng.normal(size=3)
        nrm /= np.linalg.norm(nrm)
        u = np.cross(nrm, [1.0, 0.0, 0.0] if abs(nrm[0]) < 0.9 else [0.0, 1.0, 0.0])
        u /= np.linalg.norm(u)
        v = np.cross(nrm, u)
        p = c + rng.uniform(-0.8, 0.8, (5, 1)) * u + rng.uniform(-0.8, 0.8, (5, 1)) * v + rng.normal(0, 0.01, (5, 1)) * nrm
        if i % 6 == 0:
            p[rng.integers(5)] += nrm * (0.1 + rng.uniform(-0.002, 0.002)) * rng.choice([-1, 1])
        out[i] = p.astype(np.float32)
    return out


def main():
    sets = plane_sets()
    with open(os.path.join(OUT, "planes_in.bin"), "wb") as f:
        f.write(struct.pack("<I", len(sets)))
        f.write(sets.tobytes())
    res = {}
    for o in range(4):
        po.set_eigen_order(o)
        pab = np.zeros((len(sets), 4), np.float32)
        ok = np.zeros(len(sets), np.uint32)
        for i, p in enumerate(sets):
            k, v = po.esti_plane(p)
            pab[i], ok[i] = v, int(k)
        res[f"pabcd_{po.ORDER_NAMES[o]}"] = pab
        res[f"ok_{po.ORDER_NAMES[o]}"] = ok
    po.set_eigen_order(po.ORDER_SSE)
    np.savez(os.path.join(OUT, "planes_orc.npz"), **res)

    # ---- recorded updates
    REC = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                      C.POINTER(C.c_double))
    L = po.lib()
    L.orc_set_pass_recorder.argtypes = [REC, C.c_void_p]
    cases, posts = [], []
    for seed, (M, N) in enumerate([(60000, 4000), (60000, 3000), (30000, 15), (60000, 2500)]):
        pr = synth.make_problem(M, N, "avia", cfg=200 + seed)
        m = po.Map(pr.map_xyz)
        xp, P = synth.propagate_prior_cov(po.predict, pr.x_prior)
        passes = []

        def rec(ctx, k, conv, valid, n_eff, hx, h, xa):
            hxv = np.ctypeslib.as_array(hx, (n_eff * 12,)).copy() if n_eff else np.zeros(0)
            hv = np.ctypeslib.as_array(h, (n_eff,)).copy() if n_eff else np.zeros(0)
            passes.append((valid, n_eff, hxv, hv))

        cb = REC(rec)
        L.orc_set_pass_recorder(cb, None)
        sc = po.Scan(pr.body, nthreads=1)
        x, Pn, st = sc.update_iterated(m, xp, P)
        L.orc_set_pass_recorder(C.cast(None, REC), None)
        cases.append((xp, P, 0.001, 3, passes))
        posts.append((x, Pn, st.passes))
    with open(os.path.join(OUT, "iekf_in.bin"), "wb") as f:
        f.write(struct.pack("<i", len(cases)))
        for xp, P, R, mi, passes in cases:
            f.write(np.asarray(xp, np.float64).tobytes())
            f.write(np.asarray(P, np.float64).tobytes())
            f.write(struct.pack("<dii", R, mi, len(passes)))
            for valid, n_eff, hx, h in passes:
                f.write(struct.pack("<ii", valid, n_eff))
                f.write(hx.tobytes())
                f.write(h.tobytes())
    np.savez(os.path.join(OUT, "iekf_orc.npz"), x=np.array([p[0] for p in posts]), P=np.array([p[1] for p in posts]),
             passes=np.array([p[2] for p in posts]))
    print(f"wrote {OUT}/planes_in.bin ({len(sets)} sets), iekf_in.bin ({len(cases)} updates) and the oracle's outputs")


if __name__ == "__main__":
    main()
