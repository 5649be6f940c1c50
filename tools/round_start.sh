#!/bin/bash
# First GPU call of a round (one gpurun call, ~12 minutes): the GPU suite, then the driver's bench command with the roofline
# events read both ways, then the default 300-step bench.  Everything under its own timeout; outputs in gpurun_out/round_start/.
#   gpurun --timeout 1500 -- tools/round_start.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/round_start; mkdir -p $O; cd $R
FLH_RUN_EXPERIMENTS=1 timeout 600 python -m pytest tests -q -m gpu -x -rxX 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6 | tee $O/gpu_tests.txt
for mode in deferred sync; do
  if [ $mode = sync ]; then export FLH_SYNC_EVENTS=1; else unset FLH_SYNC_EVENTS; fi
  timeout 240 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scans 0 --no-extra-legs > $O/bench20_$mode.json 2> $O/bench20_$mode.err
  echo "bench20 $mode rc=$?"; python - <<PY
import json
try:
    d = json.load(open("$O/bench20_$mode.json"))
    print({k: d[k] for k in ("value", "ms_search_pass", "ms_nosearch_pass", "device_resident_scans_per_s")}, d["roofline"]["avg_kernel_us"], d["roofline"]["fit_kernel_us"], d["config"]["event_reading"])
except Exception as e:
    print("no line:", e)
PY
done
unset FLH_SYNC_EVENTS
FLH_PLANE_CACHE=1 timeout 240 python bench.py --gpus 1 --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs 2>/dev/null | python -c "
import json, sys
try:
    d = json.loads(sys.stdin.read()); print('plane cache on :', {k: d[k] for k in ('value', 'ms_search_pass', 'ms_nosearch_pass', 'device_resident_scans_per_s')}, d['roofline']['fit_kernel_us'])
except Exception as e:
    print('plane cache run: no line', e)"
for knob in FLH_STAGE_AHEAD=2 FLH_GRAN_GROUP=32; do
  env $knob timeout 240 python bench.py --gpus 1 --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs 2>/dev/null | python -c "
import json, sys
try:
    d = json.loads(sys.stdin.read()); print('$knob :', {k: d[k] for k in ('value', 'ms_search_pass', 'ms_nosearch_pass', 'device_resident_scans_per_s')}, d['roofline']['fit_kernel_us'])
except Exception as e:
    print('$knob: no line', e)"
done
timeout 240 python bench.py --gpus 1 --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --first-stage 3 2>/dev/null | python -c "
import json, sys
try:
    d = json.loads(sys.stdin.read()); print('first_stage 3 (LDS tile):', {k: d[k] for k in ('value', 'ms_search_pass', 'ms_nosearch_pass', 'device_resident_scans_per_s')}, d['roofline']['avg_kernel_us'])
except Exception as e:
    print('first_stage 3: no line', e)"
timeout 400 python bench.py > $O/bench300.json 2> $O/bench300.err; echo "bench300 rc=$?"; cut -c1-400 $O/bench300.json; tail -3 $O/bench300.err
