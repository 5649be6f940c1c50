#!/bin/bash
# GPU box, round 4, call 7: the GPU suite on the tree with the quad summation tree, map_incremental without the host's wait and
# stream priorities; stream priorities A/B; config 2 and config 3 bench lines and kernel traces.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_call7; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
cd $R
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -60 > $O/gpu_tests.txt; tail -12 $O/gpu_tests.txt | cut -c1-200
el "gpu suite"
for v in prio noprio prio noprio; do
  L=$R/fast_lio_amd/lib/libfastlio_hip.so; [ $v = noprio ] && L=$R/fast_lio_amd/lib/libfastlio_hip_noprio.so
  FLH_LIB=$L timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs > $O/bench_$v.json 2> $O/bench_$v.err; echo "$v rc=$?"; python tools/bench_line.py $O/bench_$v.json
done
el "priority A/B"
timeout 300 python bench.py --config 3 --steps 100 --warmup 10 --scans 32 --cpu-scans 0 --no-extra-legs > $O/bench_config3.json 2> $O/bench_config3.err; echo "config 3 rc=$?"; python tools/bench_line.py $O/bench_config3.json
el "config 3"
cd /tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python $R/bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --in-process > /dev/null 2>$O/kt.err
f=$(find /tmp/kt -name '*kernel_stats.csv' 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_config2.csv && python $R/tools/kstats.py $f 6
rm -rf /tmp/kt3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -o t -- python $R/bench.py --config 3 --steps 100 --warmup 10 --scans 32 --cpu-scans 0 --no-extra-legs --in-process > /dev/null 2>$O/kt3.err
f=$(find /tmp/kt3 -name '*kernel_stats.csv' 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_config3.csv && python $R/tools/kstats.py $f 22
el "kernel traces"
exit 0
