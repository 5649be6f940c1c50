#!/bin/bash
# GPU box: round 6's evidence, ALL from one tree -- the GPU suite, the driver's command, and for every BASELINE config that fits one
# GPU (2, 3, 4, 5): the rocprofv3 kernel-trace summary, the FETCH_SIZE / WRITE_SIZE / L2 hit-miss counter passes (separate runs), and
# THEN the bench line of the same command, so that the line carries the counter traffic of this tree (bench.py reads
# profiles/rNN_pmc_summary_config*.csv and drops the figure when the sources' hash differs).  Outputs under gpurun_out/profiles_r05/.
#   echo $(git rev-parse --short HEAD) > .head_commit; gpurun --timeout 1300 -- tools/profile_round6.sh ["configs"]
CFGS=${1:-"2 3 4 5"}
RND=06
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/profiles_r$RND; mkdir -p $O
T=${STEP_TIMEOUT:-300}
export TMPDIR=/tmp
COMMIT=$(cat $R/.head_commit 2>/dev/null || echo unknown); SRC=$(python $R/tools/src_hash.py)
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
echo "commit $COMMIT sources $SRC"
args_of() { case $1 in
  2) echo "";;
  3) echo "--steps 100 --warmup 10 --scans 32";;
  4) echo "--steps 100 --warmup 10 --scans 32";;
  5) echo "--steps 60 --warmup 6 --scans 32";;
esac; }
cd $R
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -30 > $O/r${RND}_gpu_tests_tail.txt; grep -E "passed|failed|error" $O/r${RND}_gpu_tests_tail.txt | tail -3 | tee $O/r${RND}_gpu_tests.txt
el "gpu suite"
for CFG in $CFGS; do
  ARGS=$(args_of $CFG)
  # (--prelaunch 0 under the profiler: a pre-launched no-search pass sits in the queue WAITING for its state, and a trace would count
  # that wait as kernel time; with the switch off the no-search pass is the same code launched the usual way, k_fit<1,false,2>)
  PROF="--config $CFG $ARGS --cpu-scans 0 --no-extra-legs --in-process --prelaunch 0"
  cd /tmp; rm -rf /tmp/kt$CFG
  timeout $T rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$CFG -o t -- python $R/bench.py $PROF > /dev/null 2>$O/kt${CFG}.err
  rc=$?; f=$(find /tmp/kt$CFG -name '*kernel_stats.csv' 2>/dev/null | head -1)
  if [ $rc -ne 0 ] || [ -z "$f" ]; then echo "config $CFG: kernel trace failed (rc=$rc)"; tail -3 $O/kt${CFG}.err; continue; fi
  cp $f $O/r${RND}_kernel_stats_config${CFG}.csv && python $R/tools/kstats.py $f 8
  ok=1
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm${CFG}_$c
    timeout $T rocprofv3 --pmc $c --output-format csv -d /tmp/pm${CFG}_$c -o c -- python $R/bench.py $PROF --steps 40 --warmup 5 > /dev/null 2>$O/pm${CFG}_$c.err || { echo "config $CFG: counter pass $c failed"; ok=0; break; }
  done
  if [ $ok = 1 ]; then
    fa=$(find /tmp/pm${CFG}_FETCH_SIZE -name '*counter_collection.csv' | head -1); fb=$(find /tmp/pm${CFG}_WRITE_SIZE -name '*counter_collection.csv' | head -1)
    if [ -n "$fa" ] && [ -n "$fb" ]; then
      python $R/tools/pmc_summary.py $O/r${RND}_pmc_summary_config${CFG}.csv $fa $fb | grep "k_pass\|k_search\|k_fit"
      echo "{\"commit\": \"$COMMIT\", \"src_hash\": \"$SRC\", \"pass_kernel\": -1, \"command\": \"bench.py $PROF --steps 40 --warmup 5\"}" > $O/r${RND}_pmc_summary_config${CFG}.meta.json
      mkdir -p $R/profiles; cp $O/r${RND}_pmc_summary_config${CFG}.csv $O/r${RND}_pmc_summary_config${CFG}.meta.json $R/profiles/
    fi
  fi
  rm -rf /tmp/tcc$CFG; timeout $T rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/tcc$CFG -o c -- python $R/bench.py $PROF --steps 40 --warmup 5 > /dev/null 2>$O/tcc${CFG}.err
  f=$(find /tmp/tcc$CFG -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py $O/r${RND}_pmc_l2_config${CFG}.csv $f | grep "k_pass\|k_fit"
  if [ $CFG = 2 ]; then  # what bounds the kernels: issue counters and L1 line accesses (one pass each; config 2 only)
    rm -rf /tmp/sq2; timeout $T rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d /tmp/sq2 -o c -- python $R/bench.py $PROF --steps 40 --warmup 5 > /dev/null 2>$O/sq2.err
    f=$(find /tmp/sq2 -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py $O/r${RND}_pmc_sq_config2.csv $f | grep "k_pass\|k_fit"
    rm -rf /tmp/tcp2; timeout $T rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d /tmp/tcp2 -o c -- python $R/bench.py $PROF --steps 40 --warmup 5 > /dev/null 2>$O/tcp2.err
    f=$(find /tmp/tcp2 -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py $O/r${RND}_pmc_tcp_config2.csv $f | grep "k_pass\|k_fit"
  fi
  cd $R
  # the bench line LAST: it picks up the counter summary just taken
  timeout $T python bench.py --config $CFG $ARGS --cpu-scans $([ $CFG = 2 ] && echo 96 || echo 0) $([ $CFG = 2 ] || echo --no-extra-legs) > $O/r${RND}_bench_config${CFG}.json 2> $O/bench${CFG}.err
  echo "config $CFG bench rc=$?"; python tools/bench_line.py $O/r${RND}_bench_config${CFG}.json
  el "config $CFG done"
done
cd $R
timeout $T python bench.py --steps 20 --warmup 5 > $O/r${RND}_bench_driver_cmd_config2.json 2> $O/driver.err; echo "driver command rc=$?"; python tools/bench_line.py $O/r${RND}_bench_driver_cmd_config2.json
el "driver command"
# one rank, both exchanges of the sharded path (what a pass costs with the peer granules / with the RCCL all-reduce + publish kernel)
timeout $T python bench.py --steps 200 --warmup 20 --force-shard-leg --cpu-scans 0 --no-extra-legs --in-process > $O/r${RND}_bench_config2_exchanges_one_rank.json 2> $O/exch.err
python - $O/r${RND}_bench_config2_exchanges_one_rank.json <<'PY'
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
# extrinsic_est_en = 1 (the reference's default) against 0, alternating on this box
for rep in 1 2; do
  for ext in 0 1; do
    timeout $T python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --extrinsic-est $ext > $O/r${RND}_bench_config2_ext${ext}_$rep.json 2> $O/ext.err
    echo "extrinsic_est=$ext rep $rep: $(python tools/bench_line.py $O/r${RND}_bench_config2_ext${ext}_$rep.json)"
  done
done
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -2 | tee $O/r${RND}_smoke.txt
el "build + smoke"
el "all done"
exit 0
