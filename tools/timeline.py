#!/usr/bin/env python
"""Developer tool: the TIMELINE of a rocprofv3 --kernel-trace run (the *_kernel_trace.csv with one row per dispatch), read for what
the per-kernel averages hide: per pass kernel (k_pass, k_fit, k_fit_mb) its duration, the time since the previous kernel of the same
queue ended, and how long kernels of OTHER queues (the staging of the next scan on the copy stream) ran beside it.

    python tools/timeline.py <kernel_trace.csv> [label]
"""
import re
import sys

import pandas as pd

df = pd.read_csv(sys.argv[1])
label = sys.argv[2] if len(sys.argv) > 2 else ""
cols = {c.lower(): c for c in df.columns}
name_c = cols.get("kernel_name", cols.get("name"))
s_c, e_c = cols["start_timestamp"], cols["end_timestamp"]
q_c = cols.get("queue_id", cols.get("stream_id"))
pat = re.compile(r"(k_[a-z_0-9]+)")
df["k"] = [(pat.search(str(n)).group(1) if pat.search(str(n)) else str(n)[:30]) for n in df[name_c]]
df = df.sort_values(s_c).reset_index(drop=True)
t0 = df[s_c].min()
df["s"] = (df[s_c] - t0) / 1e3
df["e"] = (df[e_c] - t0) / 1e3
# drop the set-up phase: keep the last 60 % of the k_pass dispatches' time span
kp = df[df.k == "k_pass"]
if len(kp) < 8:
    sys.exit("too few k_pass dispatches")
cut = kp.s.iloc[int(len(kp) * 0.4)]
df = df[df.s >= cut].reset_index(drop=True)
main_q = df[df.k == "k_pass"][q_c].mode().iloc[0]
main = df[df[q_c] == main_q].reset_index(drop=True)
other = df[df[q_c] != main_q]
print(f"{label} queues: main {main_q} ({len(main)} dispatches), others {sorted(set(other[q_c]))} ({len(other)} dispatches: "
      f"{dict(other.k.value_counts().head(6))})")
oi = other[["s", "e"]].to_numpy()


def overlap(a, b):
    if not len(oi):
        return 0.0
    lo = oi[:, 0].clip(a, b)
    hi = oi[:, 1].clip(a, b)
    return float((hi - lo).clip(0).sum())


rows = {}
prev_e = None
prev_k = None
for _, r in main.iterrows():
    d = rows.setdefault(r.k, {"n": 0, "dur": 0.0, "gap": 0.0, "ov": 0.0, "after": {}})
    d["n"] += 1
    d["dur"] += r.e - r.s
    if prev_e is not None:
        d["gap"] += r.s - prev_e
        d["after"][prev_k] = d["after"].get(prev_k, 0) + 1
    d["ov"] += overlap(r.s, r.e)
    prev_e, prev_k = r.e, r.k
print(f"{'kernel (main queue)':22s} {'n':>5s} {'dur us':>8s} {'gap before us':>14s} {'other queues beside it us':>26s}  follows")
for k, d in sorted(rows.items(), key=lambda kv: -kv[1]["dur"]):
    n = d["n"]
    print(f"{k:22s} {n:5d} {d['dur'] / n:8.2f} {d['gap'] / n:14.2f} {d['ov'] / n:26.2f}  {d['after']}")
span = main.e.max() - main.s.min()
busy = float((main.e - main.s).sum())
print(f"main queue: span {span:.0f} us, kernels {busy:.0f} us ({100 * busy / span:.1f} %); other queues' kernels {float((other.e - other.s).sum()):.0f} us")
# do two kernels of the MAIN queue ever overlap (a pre-launched pass beside the pass before it)?
ov_main = 0
m = main.sort_values("s")
for i in range(1, len(m)):
    if m.s.iloc[i] < m.e.iloc[i - 1] - 0.05:
        ov_main += 1
print(f"main-queue dispatches that start before their predecessor has ended: {ov_main}")
