"""Developer tool: the figures of a bench.py JSON line that matter when comparing variants."""
import json
import sys

try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print({k: d.get(k) for k in ("value", "ms_search_pass", "ms_nosearch_pass", "device_resident_scans_per_s")},
          "search_us", r.get("avg_kernel_us"), "first/later", r.get("first_search_us"), r.get("later_search_us"), "fit_us", r.get("fit_kernel_us"),
          "frac", r.get("frac"))
except Exception as e:  # noqa: BLE001
    print("no line:", e)
