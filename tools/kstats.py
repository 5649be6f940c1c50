#!/usr/bin/env python
"""Developer tool: compact per-kernel table from a rocprofv3 *_kernel_stats.csv."""
import re
import sys

import pandas as pd

df = pd.read_csv(sys.argv[1])
pat = re.compile(r"(k_[a-z_0-9]+(?:<[0-9a-z, ]+>)?)")
rows = []
for _, r in df.iterrows():
    m = pat.search(r["Name"])
    name = m.group(1) if m else r["Name"][:40]
    rows.append((name, int(r["Calls"]), r["AverageNs"] / 1e3, r["MinNs"] / 1e3, r["MaxNs"] / 1e3, r["Percentage"]))
for n, c, a, mn, mx, pc in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f"{n:38s} calls={c:5d} avg={a:8.2f}us min={mn:8.2f} max={mx:8.2f} {pc:5.1f}%")
