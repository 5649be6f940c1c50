#!/bin/bash
# round 6, GPU call 38: bench.py's own host binding (--bind): off / share (the GPU's share of its node's cores, 2 CCDs here) / l3 (the
# cores of the share behind one L3, with SMT siblings) / l3nosmt, alternating on one box: the driver's command and 300-step runs.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call38; mkdir -p $O
export TMPDIR=/tmp
cd $R
line() { python - $1 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   value", d["value"], "repeats", (d.get("value_repeats") or {}).get("all"), "resident", d.get("device_resident_scans_per_s"), "search/no-search us", round(d["ms_search_pass"] * 1e3, 1), round(d["ms_nosearch_pass"] * 1e3, 1), "|", d["config"].get("host_binding"))
PY
}
for rep in 1 2 3 4; do
  for name in off share l3 l3nosmt; do
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scans 0 --no-extra-legs --bind $name > $O/bench20_${name}_$rep.json 2> $O/bench20_${name}_$rep.err
    echo "driver cmd, $name, rep $rep"; line $O/bench20_${name}_$rep.json
    timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --bind $name > $O/bench300_${name}_$rep.json 2> $O/bench300_${name}_$rep.err
    echo "300 steps, $name, rep $rep"; line $O/bench300_${name}_$rep.json
  done
done
exit 0
