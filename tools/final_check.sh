#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_final_check; mkdir -p $O; export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
cd $R
python tools/src_hash.py
timeout 1200 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | tail -4 | tee $O/gpu_tests.txt
el "GPU suite (-x)"
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -3 | tee $O/smoke.txt
el "build + smoke"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "driver command rc=$?"; python tools/bench_line.py $O/bench_driver_cmd.json
python - $O/bench_driver_cmd.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("  roofline.traffic", d["roofline"].get("traffic"), d["roofline"].get("traffic_note"), " repeats", (d.get("value_repeats") or {}).get("all"))
print("  keys", sorted(d.keys()))
PY
el "driver command"
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"; python tools/bench_line.py $O/bench_default.json
el "default bench"
exit 0
