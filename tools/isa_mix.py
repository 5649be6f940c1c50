#!/usr/bin/env python
"""Static instruction mix of the hot kernels' gfx950 code (VALU / SALU / VMEM / LDS / MFMA / waits, DPP and fp64 counts) from the
compiler's assembly -- no GPU needed.   python tools/isa_mix.py [name filter ...]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fast_lio_amd import _build  # noqa: E402

want = sys.argv[1:] or ["k_search_ring<4, 1, false, 8, false, false", "k_search_ring<16, 2", "k_fit<1, false>", "k_search_exact"]
flags = [f for f in _build.FLAGS if f not in ("-shared", "-fPIC", "-Xarch_host", "-mavx2")]


def classify(op):
    if op.startswith("v_mfma"):
        return "MFMA"
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("s_waitcnt"):
        return "WAIT"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("ds_"):
        return "LDS"
    return "other"


with tempfile.TemporaryDirectory() as td:
    asm = os.path.join(td, "k.s")
    subprocess.run([_build.hipcc()] + flags + ["-I", os.path.join(ROOT, "include"), "-x", "hip", os.path.join(_build.CSRC, "flh_kernels.hip"),
                    "--cuda-device-only", "-S", "-o", asm], check=True, capture_output=True)
    lines = open(asm).read().split("\n")
for i, l in enumerate(lines):
    if not (l.startswith("_ZN3flh") and "@_ZN3flh" in l):
        continue
    dn = subprocess.run(["c++filt", l.split(":")[0]], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(.*", "", dn).replace("void flh::", "")
    if not any(k in dn for k in want):
        continue
    c = collections.Counter()
    dpp = f64 = br = 0
    j = i + 1
    while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
        s = lines[j].strip()
        j += 1
        if not s or s.startswith((".", ";")) or s.endswith(":"):
            continue
        op = s.split()[0]
        c[classify(op)] += 1
        dpp += ("dpp" in s or "quad_perm" in s or "row_" in s)
        f64 += (op.startswith("v_") and "f64" in op)
        br += op.startswith("s_cbranch")
    print(f"{dn[:58]:58s} VALU {c['VALU']:5d} (fp64 {f64:3d}, DPP {dpp:3d})  MFMA {c['MFMA']:3d}  SALU {c['SALU']:5d}  VMEM {c['VMEM']:3d}  "
          f"LDS {c['LDS']:3d}  s_waitcnt {c['WAIT']:3d}  branches {br:3d}")
