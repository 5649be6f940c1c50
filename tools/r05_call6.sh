#!/bin/bash
# GPU box, round 5, call 6: two staging lanes / two scans staged ahead (--ring 4, the default) against one ahead (--ring 2),
# alternating; the staging and run_scans parity tests; the driver's command.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05_call6; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
cd $R
timeout 900 python -m pytest -q -m gpu tests/test_gpu_parity.py tests/test_gpu_zz_timing.py tests/test_gpu_map.py -s 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" > $O/gpu_tests_full.txt
tail -100 $O/gpu_tests_full.txt > $O/gpu_tests.txt
grep -E "passed|failed|^FAILED|^ERROR|^E  " $O/gpu_tests_full.txt | cut -c1-300 | tail -30
el "parity"
B="--steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs"
for rep in 1 2 3; do
  for v in 4 2; do
    timeout 300 python bench.py $B --ring $v > $O/bench_ring${v}_$rep.json 2> $O/bench_ring${v}_$rep.err; echo "ring $v rep $rep rc=$?"; python tools/bench_line.py $O/bench_ring${v}_$rep.json
  done
done
timeout 300 python bench.py $B --ring 4 --prelaunch 0 > $O/bench_ring4_noprelaunch.json 2> $O/bench_ring4_noprelaunch.err; echo "ring 4, prelaunch 0 rc=$?"; python tools/bench_line.py $O/bench_ring4_noprelaunch.json
el "bench A/B"
for v in 4 2; do
  timeout 400 python bench.py --steps 20 --warmup 5 --cpu-scans 0 --no-extra-legs --ring $v > $O/bench_driver_cmd_ring$v.json 2> $O/bench_driver_cmd_ring$v.err; echo "driver command, ring $v rc=$?"; python tools/bench_line.py $O/bench_driver_cmd_ring$v.json
  python - $O/bench_driver_cmd_ring$v.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  value_repeats", (d.get("value_repeats") or {}).get("all"))
except Exception as e:
    print("no line", e)
PY
done
timeout 500 python bench.py --config 3 --steps 100 --warmup 10 --scans 32 --cpu-scans 0 --no-extra-legs > $O/bench_config3.json 2> $O/bench_config3.err; echo "config 3 rc=$?"; python tools/bench_line.py $O/bench_config3.json
el "driver command, config 3"
cd /tmp; rm -rf /tmp/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $R/bench.py --steps 80 --warmup 20 --cpu-scans 0 --no-extra-legs --in-process --repeats 0 > $O/tl.json 2> $O/tl.err
f=$(find /tmp/tl -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_trace_two_lanes.csv && python $R/tools/timeline.py $f "two lanes:" | tee $O/timeline_two_lanes.txt
el "timeline"
exit 0
