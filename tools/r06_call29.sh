#!/bin/bash
# round 6, GPU call 29: the committed tree as the driver runs it, three times over: GPU suite with -x, build() + smoke(), the
# driver's command; then the default bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_final_check2; mkdir -p $O; export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
cd $R
python tools/src_hash.py
for rep in 1 2 3; do
  timeout 1200 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | tail -2 | tee $O/gpu_tests_$rep.txt
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_$rep.json 2> $O/bench_driver_cmd_$rep.err; echo "driver command $rep rc=$?: $(python tools/bench_line.py $O/bench_driver_cmd_$rep.json)"
  python - $O/bench_driver_cmd_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   repeats", (d.get("value_repeats") or {}).get("all"), "traffic", d["roofline"].get("traffic"), "attempts", d.get("measuring_process"))
PY
  el "round $rep"
done
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -3 | tee $O/smoke.txt
el "build + smoke"
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"; python tools/bench_line.py $O/bench_default.json
el "default bench"
exit 0
