#!/bin/bash
# round 6, GPU call 43: the committed tree as the driver runs it: GPU suite with -x, the driver's command twice (full line: CPU baseline
# and side legs), build() + smoke(), the default bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_final_check4; mkdir -p $O; export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
cd $R
python tools/src_hash.py
timeout 1200 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | tail -2 | tee $O/gpu_tests_1.txt
el "GPU suite"
for rep in 1 2; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_$rep.json 2> $O/bench_driver_cmd_$rep.err; echo "driver command $rep rc=$?: $(python tools/bench_line.py $O/bench_driver_cmd_$rep.json)"
  python - $O/bench_driver_cmd_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   repeats", (d.get("value_repeats") or {}).get("all"), "traffic", d["roofline"].get("traffic"), "attempts", d.get("measuring_process"))
print("   binding", d["config"].get("host_binding"), "cpu_baseline", d.get("cpu_baseline"))
print("   map_incremental", (d.get("map_incremental") or {}).get("ms_per_scan"), "keys", len(d))
PY
  el "driver command $rep"
done
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -3 | tee $O/smoke.txt
el "build + smoke"
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"; python tools/bench_line.py $O/bench_default.json
el "default bench"
exit 0
