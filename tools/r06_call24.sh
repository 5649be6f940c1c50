#!/bin/bash
# round 6, GPU call 24: a scan's first searching pass enqueued behind the previous map change (the change's counters folded in
# flh_eval_end) against the same sources built with -DFLH_SETTLE_FIRST (the counters first, as before), alternating: config 3
# (a map change after every scan), and config 2 (no map change in the timed region: must not move).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call24; mkdir -p $O
export TMPDIR=/tmp
L=$R/fast_lio_amd/lib
cd $R
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
python tools/src_hash.py
for rep in 1 2 3 4; do
  for v in old:$L/libfastlio_hip_settlefirst.so new:$L/libfastlio_hip.so; do
    IFS=: read name lib <<< "$v"
    FLH_LIB=$lib timeout 300 python bench.py --config 3 --steps 100 --warmup 10 --scans 32 --cpu-scans 0 --no-extra-legs > $O/bench_config3_${name}_$rep.json 2> $O/bench_config3_${name}_$rep.err
    echo "config 3 $name rep $rep: $(python tools/bench_line.py $O/bench_config3_${name}_$rep.json)"
    python - $O/bench_config3_${name}_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("    repeats", (d.get("value_repeats") or {}).get("all"), "mi call ms", d.get("ms_map_incremental_call_per_scan"))
PY
  done
done
el "A/B config 3"
for rep in 1 2; do
  for v in old:$L/libfastlio_hip_settlefirst.so new:$L/libfastlio_hip.so; do
    IFS=: read name lib <<< "$v"
    FLH_LIB=$lib timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs > $O/bench300_${name}_$rep.json 2> $O/bench300_${name}_$rep.err
    echo "config 2, 300 steps $name rep $rep: $(python tools/bench_line.py $O/bench300_${name}_$rep.json)"
  done
done
el "done"
exit 0
