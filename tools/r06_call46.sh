#!/bin/bash
# round 6, GPU call 46 (diagnostic, one run): the activation's wait measured where it happens (flh_scan_activate's own wait for the
# staging thread; the counter of call 35 sat behind it), five 300-step regions.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call46; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 100 python bench.py --steps 300 --warmup 30 --repeats 4 --cpu-scans 0 --no-extra-legs --diag-staging > $O/diag.json 2> $O/diag.err
python - $O/diag.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "resident", d.get("device_resident_scans_per_s"))
for r in d.get("staging_diag", []):
    print("   ", r)
PY
exit 0
