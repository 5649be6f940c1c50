#!/usr/bin/env python
"""Regenerates tests/golden/small_case.npz.

The reference (hku-mars/FAST_LIO) ships no golden vectors and cannot be built in this environment, so these
fixtures are produced by the CPU oracle (oracle/) itself: they are REGRESSION ANCHORS for the oracle and the
HIP path, not reference outputs (parity unpinned -- see oracle/fastlio_oracle.h).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fast_lio_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def main():
    pr = synth.make_problem(6000, 500, "avia", cfg=77)
    m = po.Map(pr.map_xyz)
    xp, P = synth.propagate_prior_cov(po.predict, pr.x_prior)
    out = {"map_xyz": pr.map_xyz, "body": pr.body, "x_prior": xp, "P_prior": P, "x_true": pr.x_true}
    # the fixtures under the plain keys are made with the DEFAULT summation order (ORDER_SSE, oracle_math.c); the plane
    # coefficients and flags of the other three orders ride along under o{order}_* so that every order of the device
    # code has an anchor
    for order in (0, 2, 3):
        po.set_eigen_order(order)
        sc = po.Scan(pr.body, nthreads=1)
        assert sc.h_share_model(m, xp, True, False)
        out[f"o{order}_selected"] = sc.selected
        out[f"o{order}_normvec"] = sc.normvec
    po.set_eigen_order(po.ORDER_SSE)
    out["eigen_order"] = np.int64(po.get_eigen_order())
    for ext in (0, 1):
        sc = po.Scan(pr.body, nthreads=1)
        assert sc.h_share_model(m, xp, True, bool(ext))
        HTH, HTh = sc.normal_equations()
        out[f"e{ext}_selected"] = sc.selected
        out[f"e{ext}_nn_idx"] = sc.nn_idx
        out[f"e{ext}_nn_d2"] = sc.nn_d2
        out[f"e{ext}_world"] = sc.world
        out[f"e{ext}_normvec"] = sc.normvec
        out[f"e{ext}_HTH"] = HTH
        out[f"e{ext}_HTh"] = HTh
        out[f"e{ext}_n_eff"] = np.int64(sc.n_eff)
        out[f"e{ext}_total_residual"] = np.float64(sc.total_residual)
        sc2 = po.Scan(pr.body, nthreads=1)
        x, Pn, st = sc2.update_iterated(m, xp, P, extrinsic_est_en=bool(ext))
        out[f"e{ext}_x_post"] = x
        out[f"e{ext}_P_post"] = Pn
        out[f"e{ext}_passes"] = np.int64(st.passes)
        out[f"e{ext}_searches"] = np.int64(st.searches)
        out[f"e{ext}_n_eff_per_pass"] = np.array(list(st.n_eff)[: st.passes], np.int64)
        out[f"e{ext}_selected_post"] = sc2.selected
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "small_case.npz"), **out)
    print("wrote small_case.npz:", {k: getattr(v, "shape", ()) for k, v in out.items() if k.startswith("e0")})


if __name__ == "__main__":
    main()
