#!/bin/bash
# GPU box, round 5, call 2: everything built since call 1, on one box --
#   a. the GPU suite (no -x: every failure in one go),
#   b. alternating bench pairs: HEAD / HEAD --prelaunch 0 / HEAD --index-cache 0 / round 4's HEAD (8ddc6c2) / round 4's evidence
#      tree (d015bff: the host code before the 12 x 12 information form, the arrival-order pick-up and the polling staging thread),
#   c. the driver's command twice, an extrinsic_est_en = 1 line, the config-3 stream,
#   d. tools/exchange_probe.py: a rank's share of a 1/2/4/8-way shard at 4 / 8 / 16 lanes per query, the exchanges on one rank.
# Build here first (cross-compile; the files travel with the snapshot): the product library, and the two round-4 trees under .ab/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05_call2; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
cd $R
timeout 1500 python -m pytest -q -m gpu tests -s 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | tail -120 > $O/gpu_tests.txt
grep -E "passed|failed|error|FAILED|ERROR|\[reference-sequence|\[200-step" $O/gpu_tests.txt | cut -c1-260 | tail -40
el "GPU suite"
B="--steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs"
for rep in 1 2; do
  for v in head noprelaunch noidx r4head d015bff; do
    case $v in
      head) (cd $R && timeout 300 python bench.py $B > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err);;
      noprelaunch) (cd $R && timeout 300 python bench.py $B --prelaunch 0 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err);;
      noidx) (cd $R && timeout 300 python bench.py $B --index-cache 0 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err);;
      r4head) [ -d $R/.ab/8ddc6c2 ] && (cd $R/.ab/8ddc6c2 && timeout 300 python bench.py $B --cache-dir $R/.bench_cache > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err);;
      d015bff) [ -d $R/.ab/d015bff ] && (cd $R/.ab/d015bff && timeout 300 python bench.py $B --cache-dir $R/.bench_cache > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err);;
    esac
    echo "$v $rep rc=$?"; python tools/bench_line.py $O/bench_${v}_$rep.json
  done
done
el "bench A/B"
for rep in 1 2; do
  timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd_$rep.json 2> $O/bench_driver_cmd_$rep.err; echo "driver command $rep rc=$?"; python tools/bench_line.py $O/bench_driver_cmd_$rep.json
  python - $O/bench_driver_cmd_$rep.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  value_repeats", d.get("value_repeats"), "prelaunched", d.get("prelaunched_nosearch_passes"))
    print("  map_incremental", {k: (d.get("map_incremental") or {}).get(k) for k in ("ms_per_scan", "changes")})
except Exception as e:
    print("no line", e)
PY
done
el "driver command"
timeout 300 python bench.py $B --extrinsic-est 1 > $O/bench_ext1.json 2> $O/bench_ext1.err; echo "extrinsic_est_en=1 rc=$?"; python tools/bench_line.py $O/bench_ext1.json
el "ext line"
timeout 500 python bench.py --config 3 --steps 100 --warmup 10 --scans 32 --cpu-scans 0 --no-extra-legs > $O/bench_config3.json 2> $O/bench_config3.err; echo "config 3 rc=$?"; python tools/bench_line.py $O/bench_config3.json
el "config 3"
timeout 600 python tools/exchange_probe.py > $O/exchange_probe.txt 2>&1; echo "exchange probe rc=$?"; grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" $O/exchange_probe.txt
el "exchange probe"
timeout 300 python tools/prelaunch_check.py --steps 200 > $O/prelaunch_check.txt 2>&1; echo "prelaunch check rc=$?"; tail -7 $O/prelaunch_check.txt
el "prelaunch A/B in one process"
exit 0
