#!/bin/bash
# Round 3, first GPU call: (1) GPU suite incl. the experiments that never ran, (2) the driver's bench command, (3) A/B of the
# compiled-in variants at 300 steps, (4) rocprofv3 kernel trace + SQ / TCP counter passes at HEAD, (5) a hunt for round 2's
# unexplained device fault, (6) the default bench.  Every step under its own timeout; outputs in gpurun_out/c1/.
#   gpurun --timeout 1500 -- tools/gpu_call1.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/c1; mkdir -p $O; cd $R
export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print({k: d.get(k) for k in ("value", "ms_search_pass", "ms_nosearch_pass", "device_resident_scans_per_s")}, "search_us", r.get("avg_kernel_us"), "fit_us", r.get("fit_kernel_us"))
except Exception as e:
    print("no line:", e)
PY
}
FLH_RUN_EXPERIMENTS=1 timeout 700 python -m pytest tests -q -m gpu -x -rxXs 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -25 | tee $O/gpu_tests.txt
el "gpu suite done"
timeout 240 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scans 0 --no-extra-legs > $O/bench20.json 2> $O/bench20.err; echo "bench20 rc=$?"; line $O/bench20.json
B="python bench.py --gpus 1 --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs"
timeout 240 $B > $O/v_base.json 2>$O/v_base.err; echo "base rc=$?"; line $O/v_base.json
FLH_PLANE_CACHE=1 timeout 240 $B > $O/v_plane.json 2>$O/v_plane.err; echo "plane cache rc=$?"; line $O/v_plane.json
timeout 240 $B --first-stage 3 > $O/v_tile.json 2>$O/v_tile.err; echo "tile rc=$?"; line $O/v_tile.json
FLH_STAGE_AHEAD=2 timeout 240 $B > $O/v_ahead2.json 2>$O/v_ahead2.err; echo "ahead2 rc=$?"; line $O/v_ahead2.json
timeout 240 $B --cell 2.25 > $O/v_cell225.json 2>$O/v_cell225.err; echo "cell 2.25 rc=$?"; line $O/v_cell225.json
el "variants done"
cd /tmp
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python $R/bench.py --cpu-scans 0 --no-extra-legs --in-process > /dev/null 2>$O/kt.err
echo "kernel trace rc=$?"; f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_config2.csv && python $R/tools/kstats.py $f 10
rm -rf /tmp/ktt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktt -o t -- python $R/bench.py --cpu-scans 0 --no-extra-legs --in-process --first-stage 3 > /dev/null 2>$O/ktt.err
echo "kernel trace (tile) rc=$?"; f=$(find /tmp/ktt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_config2_tile.csv && python $R/tools/kstats.py $f 6
el "traces done"
P="python $R/bench.py --steps 40 --warmup 5 --cpu-scans 0 --no-extra-legs --in-process"
rm -rf /tmp/pq; timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d /tmp/pq -o c -- $P > /dev/null 2>$O/pq.err
echo "sq pass rc=$?"; f=$(find /tmp/pq -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py $O/pmc_sq_config2.csv $f | grep "k_search\|k_fit"
rm -rf /tmp/pq2; timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d /tmp/pq2 -o c -- $P > /dev/null 2>$O/pq2.err
echo "sq pass 2 rc=$?"; f=$(find /tmp/pq2 -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py $O/pmc_sq2_config2.csv $f | grep "k_search\|k_fit"
el "counters done"
cd $R
# ---- fault hunt: the failing run of round 2 was a 300-step bench with every leg in one process
for i in 1 2 3 4; do
  timeout 200 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --in-process > $O/hunt_$i.json 2> $O/hunt_$i.err; rc=$?
  echo "hunt $i rc=$rc"; [ $rc -ne 0 ] && tail -5 $O/hunt_$i.err
done
for i in 1 2; do
  timeout 300 python bench.py --leg extras --two-streams --steps 300 > $O/hunt_extras_$i.json 2> $O/hunt_extras_$i.err; rc=$?
  echo "hunt extras $i rc=$rc"; [ $rc -ne 0 ] && tail -5 $O/hunt_extras_$i.err
done
el "hunt done"
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver bench rc=$?"; cut -c1-1200 $O/bench_driver.json
el "all done"
