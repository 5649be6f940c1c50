#!/usr/bin/env python
"""Per-kernel means of rocprofv3 --pmc counters.

usage: pmc_summary.py OUT.csv counter_collection.csv [counter_collection.csv ...]

Each input is one --pmc pass (the TCC block cannot hold FETCH_SIZE and WRITE_SIZE in one pass,
/opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots").  Output: kernel, counter, dispatches, mean, sum.
"""
import re
import sys

import pandas as pd

pat = re.compile(r"(k_[a-z_0-9]+(?:<[0-9a-z, ]+>)?)")


def short(n):
    m = pat.search(n)
    return m.group(1) if m else n[:48]


frames = []
for f in sys.argv[2:]:
    df = pd.read_csv(f, usecols=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
    df["kernel"] = df["Kernel_Name"].map(short)
    # a counter can be reported per XCD/instance: sum within a dispatch first
    d = df.groupby(["kernel", "Counter_Name", "Dispatch_Id"], as_index=False)["Counter_Value"].sum()
    g = d.groupby(["kernel", "Counter_Name"])["Counter_Value"].agg(["count", "mean", "sum"]).reset_index()
    frames.append(g)
out = pd.concat(frames).rename(columns={"Counter_Name": "counter", "count": "dispatches"})
out.to_csv(sys.argv[1], index=False, float_format="%.6g")
print(out.to_string(index=False))
