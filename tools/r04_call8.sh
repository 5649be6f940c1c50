#!/bin/bash
# GPU box, round 4, call 8: map tests after the far-nearest restructure and the enqueued map change; reversed dispatch order of k_pass
# A/B (bench's own kernel events + value, same box); config 3 bench + kernel trace.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_call8; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
cd $R
timeout 600 python -m pytest -q -m gpu tests/test_gpu_map.py tests/test_gpu_zz_timing.py tests/test_gpu_parity.py 2>&1 | tail -40 > $O/gpu_tests.txt; tail -8 $O/gpu_tests.txt | cut -c1-200
el "tests"
for v in base rev base rev; do
  L=$R/fast_lio_amd/lib/libfastlio_hip.so; [ $v = rev ] && L=$R/fast_lio_amd/lib/libfastlio_hip_rev.so
  FLH_LIB=$L timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --timing-samples 64 > $O/bench_$v.json 2> $O/bench_$v.err; echo "$v rc=$?"; python tools/bench_line.py $O/bench_$v.json
done
el "reversed dispatch A/B"
timeout 300 python bench.py --config 3 --steps 100 --warmup 10 --scans 32 --cpu-scans 0 --no-extra-legs > $O/bench_config3.json 2> $O/bench_config3.err; echo "config 3 rc=$?"; python tools/bench_line.py $O/bench_config3.json
cd /tmp; rm -rf /tmp/kt3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -o t -- python $R/bench.py --config 3 --steps 100 --warmup 10 --scans 32 --cpu-scans 0 --no-extra-legs --in-process > /dev/null 2>$O/kt3.err
f=$(find /tmp/kt3 -name '*kernel_stats.csv' 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_config3.csv && python $R/tools/kstats.py $f 16
el "config 3"
exit 0
