#!/bin/bash
# GPU box, round 4, call 9: the map tests after the list-based far-nearest search; config 3 bench + kernel trace.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_call9; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
cd $R
timeout 600 python -m pytest -q -m gpu tests/test_gpu_map.py tests/test_gpu_zz_timing.py 2>&1 | tail -30 > $O/gpu_tests.txt; tail -6 $O/gpu_tests.txt | cut -c1-200
el "tests"
timeout 300 python bench.py --config 3 --steps 100 --warmup 10 --scans 32 --cpu-scans 0 --no-extra-legs > $O/bench_config3.json 2> $O/bench_config3.err; echo "config 3 rc=$?"; python tools/bench_line.py $O/bench_config3.json
cd /tmp; rm -rf /tmp/kt3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -o t -- python $R/bench.py --config 3 --steps 100 --warmup 10 --scans 32 --cpu-scans 0 --no-extra-legs --in-process > /dev/null 2>$O/kt3.err
f=$(find /tmp/kt3 -name '*kernel_stats.csv' 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_config3.csv && python $R/tools/kstats.py $f 18
el "config 3"
exit 0
