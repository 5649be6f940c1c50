#!/bin/bash
# round 6, GPU call 10: config 4 (130 000-point scans: 32 tiles) with the library's own staging against the vendor sort, alternating;
# the driver's command three times (the evidence run's first region was an outlier: 5 202 against repeats of 7 123-7 370).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call10; mkdir -p $O
export TMPDIR=/tmp
cd $R
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
for rep in 1 2; do
  for ss in 0 1; do
    timeout 400 python bench.py --config 4 --steps 100 --warmup 10 --scans 32 --cpu-scans 0 --no-extra-legs --stage-sort $ss > $O/bench_config4_ss${ss}_$rep.json 2> $O/bench_config4_ss${ss}_$rep.err
    echo "config 4 stage_sort=$ss rep $rep: $(python tools/bench_line.py $O/bench_config4_ss${ss}_$rep.json)"
  done
done
el "config 4 pairs"
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd_$rep.json 2> $O/bench_driver_cmd_$rep.err
  echo "driver command $rep: $(python tools/bench_line.py $O/bench_driver_cmd_$rep.json)"
  python - $O/bench_driver_cmd_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   repeats", (d.get("value_repeats") or {}).get("all"), "traffic", d["roofline"].get("traffic"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
done
el "driver command x3"
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -3 | tee $O/smoke.txt
el "build + smoke"
exit 0
