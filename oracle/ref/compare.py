#!/usr/bin/env python
"""Compares the outputs of the oracle/_ref binaries (the reference's own code) with the oracle's (make_inputs.py)."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "oracle", "_ref")


def compare_planes():
    raw = open(os.path.join(OUT, "planes_ref.bin"), "rb").read()
    n = struct.unpack_from("<I", raw)[0]
    rec = np.frombuffer(raw, dtype=np.uint32, offset=4).reshape(n, 5)
    ref_bits, ref_ok = rec[:, :4], rec[:, 4]
    orc = np.load(os.path.join(OUT, "planes_orc.npz"))
    result = {}
    for name in ("seq", "sse", "pairwise", "novec"):
        bits = orc[f"pabcd_{name}"].view(np.uint32)
        fin = np.isfinite(orc[f"pabcd_{name}"]).all(axis=1)
        same = (bits == ref_bits).all(axis=1) | ~fin
        result[name] = (int((~same).sum()), int((orc[f"ok_{name}"] != ref_ok).sum()))
        print(f"order {name:9s}: {result[name][0]:5d} of {n} fits differ in pabcd bits, {result[name][1]} verdicts differ")
    best = min(result, key=lambda k: result[k])
    print("matching order:", best if result[best] == (0, 0) else f"NONE exactly (closest: {best})")
    return result


def compare_iekf():
    raw = open(os.path.join(OUT, "iekf_ref.bin"), "rb").read()
    nc = struct.unpack_from("<i", raw)[0]
    off = 4
    orc = np.load(os.path.join(OUT, "iekf_orc.npz"))
    worst_x = worst_p = 0.0
    for c in range(nc):
        used = struct.unpack_from("<i", raw, off)[0]
        off += 4
        x = np.frombuffer(raw, np.float64, 26, off)
        off += 26 * 8
        P = np.frombuffer(raw, np.float64, 529, off).reshape(23, 23)
        off += 529 * 8
        assert used == int(orc["passes"][c]), f"case {c}: the real filter used {used} passes, the oracle {int(orc['passes'][c])}"
        worst_x = max(worst_x, float(np.abs(x - orc["x"][c]).max()))
        worst_p = max(worst_p, float(np.abs(P - orc["P"][c]).max() / np.abs(orc["P"][c]).max()))
    print(f"IEKF: {nc} updates, max |x_ref - x_oracle| = {worst_x:.3e}, max |P_ref - P_oracle| / max|P| = {worst_p:.3e}")
    return worst_x, worst_p


if __name__ == "__main__":
    ok = True
    if os.path.exists(os.path.join(OUT, "planes_ref.bin")):
        r = compare_planes()
        ok &= r["sse"] == (0, 0)
    if os.path.exists(os.path.join(OUT, "iekf_ref.bin")):
        wx, wp = compare_iekf()
        ok &= wx < 1e-9 and wp < 1e-9
    sys.exit(0 if ok else 1)
