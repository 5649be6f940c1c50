#!/bin/bash
# GPU box: the round's evidence for one BASELINE config -- the bench line, the rocprofv3 kernel-trace summary of the same
# command, and (PMC=1) the FETCH_SIZE / WRITE_SIZE passes.  Outputs under gpurun_out/profiles_r02/ (copy into profiles/).
#   tools/profile_round.sh CONFIG "bench args" [PMC]
CFG=$1; ARGS=$2; PMC=${3:-0}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py --config $CFG $ARGS > $O/r02_bench_config${CFG}.json 2> $O/r02_bench_config${CFG}.err
echo "bench rc=$?"; cut -c1-600 $O/r02_bench_config${CFG}.json
rm -rf /tmp/kt$CFG; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$CFG -o t -- python $R/bench.py --config $CFG $ARGS --cpu-scans 0 --no-extra-legs --in-process > /dev/null 2>&1
f=$(find /tmp/kt$CFG -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r02_kernel_stats_config${CFG}.csv && python $R/tools/kstats.py $f 8
if [ "$PMC" = "1" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm${CFG}_$c; timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/pm${CFG}_$c -o c -- python $R/bench.py --config $CFG $ARGS --steps 40 --warmup 5 --cpu-scans 0 --no-extra-legs --in-process > /dev/null 2>&1
  done
  fa=$(find /tmp/pm${CFG}_FETCH_SIZE -name '*counter_collection.csv' | head -1); fb=$(find /tmp/pm${CFG}_WRITE_SIZE -name '*counter_collection.csv' | head -1)
  [ -n "$fa" ] && [ -n "$fb" ] && python $R/tools/pmc_summary.py $O/r02_pmc_summary_config${CFG}.csv $fa $fb | grep "k_search\|k_fit" 
fi
