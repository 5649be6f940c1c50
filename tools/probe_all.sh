#!/bin/bash
# Developer tool (GPU box): per-kernel durations (and optionally PMC counters) of the search stage for several variants.
# usage: tools/probe_all.sh OUTDIR "lpq:stage ..." "PMC counters or empty" [extra search_probe.py args]
OUT=$1; shift
VARS=$1; shift
PMC=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  lpq=${v%%:*}; st=${v##*:}
  tag=l${lpq}s${st}
  timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$tag -o t -- python $R/tools/search_probe.py --lpq $lpq --stage $st "$@" > /dev/null 2>&1
  f=$(find /tmp/pp_$tag -name '*kernel_stats.csv' | head -1)
  echo "== $tag" >> $R/$OUT/kstats.txt
  python $R/tools/kstats.py $f 5 | grep "k_search\|k_fit\|k_update" >> $R/$OUT/kstats.txt
  cp $f $R/$OUT/${tag}_kernel_stats.csv
  if [ -n "$PMC" ]; then
    timeout 150 rocprofv3 --pmc $PMC --output-format csv -d /tmp/pq_$tag -o c -- python $R/tools/search_probe.py --lpq $lpq --stage $st --reps 2 "$@" > /tmp/pq_$tag.log 2>&1
    echo "rc=$?" >> /tmp/pq_$tag.log
    f=$(find /tmp/pq_$tag -name '*counter_collection.csv' | head -1)
    echo "== $tag" >> $R/$OUT/pmc.txt
    if [ -n "$f" ]; then python $R/tools/pmc_summary.py $R/$OUT/${tag}_pmc.csv $f | grep -i "k_search\|k_fit\|k_update\|kernel" >> $R/$OUT/pmc.txt; else tail -5 /tmp/pq_$tag.log >> $R/$OUT/pmc.txt; fi
  fi
done
cat $R/$OUT/kstats.txt
[ -n "$PMC" ] && cat $R/$OUT/pmc.txt
