#!/bin/bash
# Round 3, third GPU call: GPU suite at HEAD (wave-tile v2, deferred map bookkeeping), first-stage variants (bench + kernel trace +
# SQ counters), the side legs (map_incremental timing) and a config-3 line.  Outputs in gpurun_out/c3/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/c3; mkdir -p $O; cd $R
export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
line() { python $R/tools/bench_line.py "$1"; }
timeout 900 python -m pytest tests -q -m gpu -x -rxXs 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -25 | tee $O/gpu_tests.txt
el "gpu suite done"
B="python bench.py --gpus 1 --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs"
for fs in 1 3 4; do
  timeout 300 $B --first-stage $fs > $O/v_fs$fs.json 2>$O/v_fs$fs.err; echo "first_stage $fs rc=$?"; line $O/v_fs$fs.json
done
el "variants done"
cd /tmp
for fs in 3 4; do
  rm -rf /tmp/kt$fs; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$fs -o t -- python $R/bench.py --steps 150 --warmup 20 --scans 100 --cpu-scans 0 --no-extra-legs --in-process --first-stage $fs > /dev/null 2>$O/kt$fs.err
  echo "kernel trace fs=$fs rc=$?"; f=$(find /tmp/kt$fs -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_fs$fs.csv && python $R/tools/kstats.py $f 5
  rm -rf /tmp/pq$fs; timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d /tmp/pq$fs -o c -- python $R/bench.py --steps 40 --warmup 5 --scans 40 --cpu-scans 0 --no-extra-legs --in-process --first-stage $fs > /dev/null 2>$O/pq$fs.err
  echo "sq pass fs=$fs rc=$?"; f=$(find /tmp/pq$fs -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py $O/pmc_sq_fs$fs.csv $f | grep "k_search_wtile"
done
el "traces + counters done"
cd $R
timeout 300 python bench.py --leg extras --steps 40 > $O/extras.json 2> $O/extras.err; echo "extras rc=$?"; cut -c1-900 $O/extras.json
timeout 400 python bench.py --config 3 --steps 60 --warmup 10 --scans 24 --cpu-scans 0 --no-extra-legs > $O/bench_config3.json 2> $O/bench_config3.err; echo "config 3 rc=$?"; line $O/bench_config3.json; python - <<PY
import json
try:
    d = json.loads(open("$O/bench_config3.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("ms_per_step", "ms_map_incremental_call_per_scan", "passes_per_scan", "searches_per_scan")})
except Exception as e:
    print("no config-3 line", e)
PY
el "all done"
