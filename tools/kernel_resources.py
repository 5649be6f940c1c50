#!/usr/bin/env python
"""Registers, LDS, scratch and occupancy of the hot kernels as the compiler reports them for gfx950
(hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).   python tools/kernel_resources.py [filter ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fast_lio_amd import _build  # noqa: E402

want = sys.argv[1:] or ["k_search", "k_fit", "k_fill_d2", "k_scan_restride", "k_publish", "k_mi_classify", "k_brick_rewrite", "k_undistort"]
flags = [f for f in _build.FLAGS if f not in ("-shared", "-fPIC")]
with tempfile.TemporaryDirectory() as td:
    for src in _build.SOURCES:
        if not src.endswith(".hip"):
            continue
        r = subprocess.run([_build.hipcc()] + flags + ["-I", os.path.join(ROOT, "include"), "-c", "-x", "hip", os.path.join(_build.CSRC, src),
                            "-o", os.path.join(td, "o.o"), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        seen = set()
        for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
            name = b.split("\n")[0].split(" ")[0].strip()
            if name in seen:
                continue
            seen.add(name)
            nm = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            nm = re.sub(r"\(.*", "", nm).replace("void flh::", "")
            if not any(k in nm for k in want):
                continue

            def g(k):
                m = re.search(k + r": (\d+)", b)
                return int(m.group(1)) if m else -1

            scratch, lds, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]")
            print(f"{nm[:60]:60s} VGPR {g('VGPRs'):3d} AGPR {g('AGPRs'):2d} SGPR {g('SGPRs'):3d} scratch {scratch:3d} "
                  f"LDS {lds:6d} B/block  occupancy {occ} waves/SIMD")
