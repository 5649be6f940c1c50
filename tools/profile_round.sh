#!/bin/bash
# GPU box: the round's evidence for one BASELINE config -- the bench line, the rocprofv3 kernel-trace summary of the same
# command, and (PMC=1) the FETCH_SIZE / WRITE_SIZE passes.  Outputs under gpurun_out/profiles_rNN/ (copy into profiles/).
#   tools/profile_round.sh CONFIG "bench args" [PMC] [ROUND]
# Every step runs under its own short timeout, and a step that fails ends the script: a bench that dies must not be
# followed by three profiler runs of the same command that each wait for their limit (that cost round 2 half an hour).
CFG=$1; ARGS=$2; PMC=${3:-0}; RND=${4:-03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/profiles_r$RND
T=${STEP_TIMEOUT:-420}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout $T python $R/bench.py --config $CFG $ARGS > $O/r${RND}_bench_config${CFG}.json 2> $O/r${RND}_bench_config${CFG}.err
rc=$?; echo "bench rc=$rc"; cut -c1-600 $O/r${RND}_bench_config${CFG}.json
if [ $rc -ne 0 ] || [ ! -s $O/r${RND}_bench_config${CFG}.json ]; then tail -5 $O/r${RND}_bench_config${CFG}.err; echo "bench failed: profiler runs skipped"; exit 1; fi
rm -rf /tmp/kt$CFG
timeout $T rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$CFG -o t -- python $R/bench.py --config $CFG $ARGS --cpu-scans 0 --no-extra-legs --in-process > /dev/null 2>&1
rc=$?; echo "kernel trace rc=$rc"
f=$(find /tmp/kt$CFG -name '*kernel_stats.csv' 2>/dev/null | head -1)
if [ $rc -ne 0 ] || [ -z "$f" ]; then echo "kernel trace failed: counter passes skipped"; exit 1; fi
cp $f $O/r${RND}_kernel_stats_config${CFG}.csv && python $R/tools/kstats.py $f 8
if [ "$PMC" = "1" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm${CFG}_$c
    timeout $T rocprofv3 --pmc $c --output-format csv -d /tmp/pm${CFG}_$c -o c -- python $R/bench.py --config $CFG $ARGS --steps 40 --warmup 5 --cpu-scans 0 --no-extra-legs --in-process > /dev/null 2>&1 || { echo "counter pass $c failed"; exit 1; }
  done
  fa=$(find /tmp/pm${CFG}_FETCH_SIZE -name '*counter_collection.csv' | head -1); fb=$(find /tmp/pm${CFG}_WRITE_SIZE -name '*counter_collection.csv' | head -1)
  [ -n "$fa" ] && [ -n "$fb" ] && python $R/tools/pmc_summary.py $O/r${RND}_pmc_summary_config${CFG}.csv $fa $fb | grep "k_search\|k_fit"
fi
exit 0
