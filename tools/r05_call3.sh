#!/bin/bash
# GPU box, round 5, call 3: the fixes after call 2 on one box -- the GPU suite (no -x), alternating bench pairs HEAD / HEAD
# --prelaunch 0 / round 4's HEAD (8ddc6c2), the driver's command twice (the idle gap between warm-up and measurement is gone),
# the config-3 stream, the one-rank exchanges (peer granules / RCCL on group totals) in the native loop, tools/exchange_probe.py.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05_call3; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
cd $R
timeout 1500 python -m pytest -q -m gpu tests -s 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" > $O/gpu_tests_full.txt
tail -150 $O/gpu_tests_full.txt > $O/gpu_tests.txt
grep -E "passed|failed|^FAILED|^ERROR|\[reference-sequence|\[200-step" $O/gpu_tests_full.txt | cut -c1-300 | tail -40
el "GPU suite"
B="--steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs"
for rep in 1 2; do
  for v in head noprelaunch r4head; do
    case $v in
      head) (cd $R && timeout 300 python bench.py $B > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err);;
      noprelaunch) (cd $R && timeout 300 python bench.py $B --prelaunch 0 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err);;
      r4head) [ -d $R/.ab/8ddc6c2 ] && (cd $R/.ab/8ddc6c2 && timeout 300 python bench.py $B --cache-dir $R/.bench_cache > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err);;
    esac
    echo "$v $rep rc=$?"; python tools/bench_line.py $O/bench_${v}_$rep.json
  done
done
python - $O/bench_head_1.json $O/bench_head_2.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("  prelaunched", d.get("prelaunched_nosearch_passes"), "value_repeats", (d.get("value_repeats") or {}).get("all"))
    except Exception as e:
        print("no line", e)
PY
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
(cd $R/.ab/8ddc6c2 && timeout 400 python bench.py --steps 20 --warmup 5 --cpu-scans 0 --no-extra-legs --cache-dir $R/.bench_cache > $O/bench_driver_cmd_r4head.json 2> $O/bench_driver_cmd_r4head.err); echo "driver command, round 4's head rc=$?"; python tools/bench_line.py $O/bench_driver_cmd_r4head.json
el "driver command"
timeout 500 python bench.py --config 3 --steps 100 --warmup 10 --scans 32 --cpu-scans 0 --no-extra-legs > $O/bench_config3.json 2> $O/bench_config3.err; echo "config 3 rc=$?"; python tools/bench_line.py $O/bench_config3.json
el "config 3"
timeout 300 python bench.py --steps 200 --warmup 20 --force-shard-leg --cpu-scans 0 --no-extra-legs --in-process > $O/bench_config2_exchanges_one_rank.json 2> $O/exch.err
python - $O/bench_config2_exchanges_one_rank.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    for k in ("shard_mode", "other_exchange"):
        e = d.get(k) or {}
        print(k, {x: e.get(x) for x in ("value", "ms_search_pass", "ms_nosearch_pass", "ranks_in_communicator", "error")}, (e.get("collective") or "")[:40])
    print("plain", {x: d.get(x) for x in ("value", "ms_search_pass", "ms_nosearch_pass")})
except Exception as e:
    print("no line", e)
PY
el "one-rank exchanges"
timeout 600 python tools/exchange_probe.py > $O/exchange_probe.txt 2>&1; echo "exchange probe rc=$?"; grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" $O/exchange_probe.txt
el "exchange probe"
timeout 300 python tools/prelaunch_check.py --steps 200 > $O/prelaunch_check.txt 2>&1; echo "prelaunch check rc=$?"; tail -7 $O/prelaunch_check.txt
el "prelaunch A/B in one process"
exit 0
