#!/bin/bash
# GPU box: the round's evidence, ALL from one tree -- for every BASELINE config that fits one GPU (2, 3, 4, 5) the bench line, the
# rocprofv3 kernel-trace summary of the same command and the FETCH_SIZE / WRITE_SIZE counter passes; the driver's command; the
# fp32-vs-fp16 plane-fit ablation on config 5.  Outputs under gpurun_out/profiles_rNN/ (copy into profiles/).
#   echo $(git rev-parse --short HEAD) > .head_commit; gpurun --timeout 2400 -- tools/profile_round.sh [ROUND] ["configs"]
# Every step runs under its own timeout; a config whose bench fails is skipped (no profiler run waits for a dead command).
RND=${1:-03}; CFGS=${2:-"2 3 4 5"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/profiles_r$RND; mkdir -p $O
T=${STEP_TIMEOUT:-420}
export TMPDIR=/tmp
COMMIT=$(cat $R/.head_commit 2>/dev/null || echo unknown); SRC=$(python $R/tools/src_hash.py)
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
echo "commit $COMMIT sources $SRC"
args_of() { case $1 in
  2) echo "";;
  3) echo "--steps 100 --warmup 10 --scans 32";;
  4) echo "--steps 100 --warmup 10 --scans 8";;
  5) echo "--steps 60 --warmup 6 --scans 8";;
esac; }
cd $R
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/r${RND}_gpu_tests.txt
el "gpu suite"
timeout $T python bench.py --steps 20 --warmup 5 > $O/r${RND}_bench_driver_cmd_config2.json 2> $O/driver.err; echo "driver command rc=$?"; python tools/bench_line.py $O/r${RND}_bench_driver_cmd_config2.json
el "driver command"
for CFG in $CFGS; do
  ARGS=$(args_of $CFG)
  cd $R
  timeout $T python bench.py --config $CFG $ARGS --cpu-scans $([ $CFG = 2 ] && echo 96 || echo 0) > $O/r${RND}_bench_config${CFG}.json 2> $O/bench${CFG}.err
  rc=$?; echo "config $CFG bench rc=$rc"; python tools/bench_line.py $O/r${RND}_bench_config${CFG}.json
  if [ $rc -ne 0 ] || [ ! -s $O/r${RND}_bench_config${CFG}.json ]; then tail -5 $O/bench${CFG}.err; echo "config $CFG: bench failed, profiler runs skipped"; continue; fi
  cd /tmp; rm -rf /tmp/kt$CFG
  timeout $T rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$CFG -o t -- python $R/bench.py --config $CFG $ARGS --cpu-scans 0 --no-extra-legs --in-process > /dev/null 2>$O/kt${CFG}.err
  rc=$?; f=$(find /tmp/kt$CFG -name '*kernel_stats.csv' 2>/dev/null | head -1)
  if [ $rc -ne 0 ] || [ -z "$f" ]; then echo "config $CFG: kernel trace failed (rc=$rc), counter passes skipped"; continue; fi
  cp $f $O/r${RND}_kernel_stats_config${CFG}.csv && python $R/tools/kstats.py $f 8
  ok=1
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm${CFG}_$c
    timeout $T rocprofv3 --pmc $c --output-format csv -d /tmp/pm${CFG}_$c -o c -- python $R/bench.py --config $CFG $ARGS --steps 40 --warmup 5 --cpu-scans 0 --no-extra-legs --in-process > /dev/null 2>$O/pm${CFG}_$c.err || { echo "config $CFG: counter pass $c failed"; ok=0; break; }
  done
  if [ $ok = 1 ]; then
    fa=$(find /tmp/pm${CFG}_FETCH_SIZE -name '*counter_collection.csv' | head -1); fb=$(find /tmp/pm${CFG}_WRITE_SIZE -name '*counter_collection.csv' | head -1)
    if [ -n "$fa" ] && [ -n "$fb" ]; then
      python $R/tools/pmc_summary.py $O/r${RND}_pmc_summary_config${CFG}.csv $fa $fb | grep "k_pass\|k_search\|k_fit"
      echo "{\"commit\": \"$COMMIT\", \"src_hash\": \"$SRC\", \"pass_kernel\": -1, \"command\": \"bench.py --config $CFG $ARGS --steps 40 --warmup 5 --cpu-scans 0 --no-extra-legs --in-process\"}" > $O/r${RND}_pmc_summary_config${CFG}.meta.json
    fi
  fi
  if [ $CFG = 2 ]; then  # what bounds the kernels: issue counters and L1 line accesses (one pass each; config 2 only)
    rm -rf /tmp/sq2; timeout $T rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d /tmp/sq2 -o c -- python $R/bench.py --config 2 --steps 40 --warmup 5 --cpu-scans 0 --no-extra-legs --in-process > /dev/null 2>$O/sq2.err
    f=$(find /tmp/sq2 -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py $O/r${RND}_pmc_sq_config2.csv $f | grep "k_pass\|k_search\|k_fit"
    rm -rf /tmp/tcp2; timeout $T rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d /tmp/tcp2 -o c -- python $R/bench.py --config 2 --steps 40 --warmup 5 --cpu-scans 0 --no-extra-legs --in-process > /dev/null 2>$O/tcp2.err
    f=$(find /tmp/tcp2 -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py $O/r${RND}_pmc_tcp_config2.csv $f | grep "k_pass\|k_search\|k_fit"
  fi
  el "config $CFG done"
done
# ---- BASELINE configs[4]'s ablation: the plane fit in fp32 vs fp16 (plane cache off for both, so that every pass fits), kernel trace
for dt in 0 1; do
  cd /tmp; rm -rf /tmp/kf$dt
  timeout $T rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kf$dt -o t -- python $R/bench.py --config 5 $(args_of 5) --cpu-scans 0 --no-extra-legs --in-process --plane-cache 0 --plane-fit-dtype $dt > $O/r${RND}_bench_config5_planefit_dtype$dt.json 2>$O/kf$dt.err
  echo "ablation dtype=$dt rc=$?"; f=$(find /tmp/kf$dt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r${RND}_kernel_stats_config5_planefit_dtype$dt.csv && python $R/tools/kstats.py $f 12 | grep "k_fit"
  python $R/tools/bench_line.py $O/r${RND}_bench_config5_planefit_dtype$dt.json
done
el "all done"
exit 0
