#!/bin/bash
# GPU box, round 4, call 14: the driver's command after the map_incremental side leg began to call the library as the node does.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_call14; mkdir -p $O
cd $R
timeout 150 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/driver.err; echo "driver command rc=$? stdout lines: $(wc -l < $O/bench_driver_cmd.json)"; python tools/bench_line.py $O/bench_driver_cmd.json
python - $O/bench_driver_cmd.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d.get("map_incremental")); print(d.get("measuring_process")); print({k: d.get(k) for k in ("extras_error",)})
PY
exit 0
