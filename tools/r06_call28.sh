#!/bin/bash
# round 6, GPU call 28 (diagnostic): why is the contract's region of the driver's command (20 scans nobody has seen, right behind a
# five-scan warm-up) slower than its own repeats?  (a) as is; (b) the repeats on scans nobody has seen either; (c) every scan's host
# buffer crossed PCIe once before the warm-up; (d) every scan searched once against the map before the warm-up.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call28; mkdir -p $O
export TMPDIR=/tmp
cd $R
show() { python - $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("   value %.0f repeats %s resident %.0f  passes %.1f / %.1f us  k_pass first/later %.1f / %.1f" % (d["value"], d["value_repeats"]["all"][1:], d.get("device_resident_scans_per_s") or 0, d["ms_search_pass"] * 1e3, d["ms_nosearch_pass"] * 1e3, r.get("first_search_us") or 0, r.get("later_search_us") or 0))
except Exception as e:
    print("   no line", e)
PY
}
for rep in 1 2; do
  for v in "a:" "b:--diag-fresh-repeats" "c:--diag-pretouch 1 --diag-fresh-repeats" "d:--diag-pretouch 2 --diag-fresh-repeats"; do
    name=${v%%:*}; flags=${v#*:}
    timeout 300 python bench.py --steps 20 --warmup 5 --cpu-scans 0 --no-extra-legs $flags > $O/bench20_${name}_$rep.json 2> $O/bench20_${name}_$rep.err
    echo "($name) $flags rep $rep"; show $O/bench20_${name}_$rep.json
  done
done
exit 0
