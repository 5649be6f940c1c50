#!/bin/bash
# round 6, GPU call 23: a scan's first searching pass enqueued behind the previous map change (the change's counters folded in flh_eval_end), against the library of 2959b62; same layout as calls 17-20.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${CALL:-r06_call23}; mkdir -p $O
export TMPDIR=/tmp
L=$R/fast_lio_amd/lib
cd $R
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
python tools/src_hash.py
timeout 900 python -m pytest tests/test_gpu_map.py -q -m gpu -x > $O/map_tests.txt 2>&1; grep -E "passed|failed|error|Error|assert" $O/map_tests.txt | head -20
el "map tests"
timeout 1500 python -m pytest tests/ -q -m gpu -x > $O/gpu_tests.txt 2>&1; grep -E "passed|failed|error|Error|assert" $O/gpu_tests.txt | head -20
el "gpu suite"
for rep in 1 2 3; do
  for v in old:$L/libfastlio_hip_2959.so new:$L/libfastlio_hip.so; do
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
for v in old:$L/libfastlio_hip_2959.so new:$L/libfastlio_hip.so; do
  IFS=: read name lib <<< "$v"
  cd /tmp; rm -rf /tmp/kt3
  FLH_LIB=$lib timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -o t -- python $R/bench.py --config 3 --steps 100 --warmup 10 --scans 32 --cpu-scans 0 --no-extra-legs --in-process --prelaunch 0 > /dev/null 2> $O/kt3_$name.err
  f=$(find /tmp/kt3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_config3_$name.csv && echo "== $name" && python $R/tools/kstats.py $f 14
  cd $R
done
el "kernel stats"
for rep in 1 2; do
  for v in old:$L/libfastlio_hip_2959.so new:$L/libfastlio_hip.so; do
    IFS=: read name lib <<< "$v"
    FLH_LIB=$lib timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 > $O/bench300_${name}_$rep.json 2> $O/bench300_${name}_$rep.err
    echo "config 2, 300 steps $name rep $rep: $(python tools/bench_line.py $O/bench300_${name}_$rep.json)"
    python - $O/bench300_${name}_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("    map_incremental side leg:", {k: v for k, v in d.items() if "map_incremental" in k})
PY
  done
done
el "done"
exit 0
