#!/bin/bash
# GPU box, round 4, call 4 (the results of calls 1-3 were lost with the container): the GPU suite, the driver's command, the
# one-launch pass against the three-launch pass on the same box, the kernel trace of both, the phase stamps of k_pass.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_call4; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
cd $R
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -40 > $O/gpu_tests.txt; tail -15 $O/gpu_tests.txt
el "gpu suite"
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/driver.err; echo "driver command rc=$?"; tail -3 $O/driver.err; python tools/bench_line.py $O/bench_driver_cmd.json
el "driver command"
for pk in 1 0; do
  timeout 300 python bench.py --steps 300 --warmup 30 --pass-kernel $pk --cpu-scans 0 --no-extra-legs > $O/bench_pk$pk.json 2> $O/bench_pk$pk.err; echo "pass-kernel $pk rc=$?"; python tools/bench_line.py $O/bench_pk$pk.json
done
el "A/B"
cd /tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python $R/bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --in-process > /dev/null 2>$O/kt.err
f=$(find /tmp/kt -name '*kernel_stats.csv' 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_config2.csv && python $R/tools/kstats.py $f 12
el "kernel trace"
cd /tmp; rm -rf /tmp/kt0
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt0 -o t -- python $R/bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --in-process --pass-kernel 0 > /dev/null 2>$O/kt0.err
f=$(find /tmp/kt0 -name '*kernel_stats.csv' 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_config2_three_launch.csv && python $R/tools/kstats.py $f 8
el "kernel trace (three-launch)"
cd $R
FLH_LIB=$R/fast_lio_amd/lib/libfastlio_hip_stamps.so timeout 300 python tools/pass_stamps.py > $O/pass_stamps.txt 2>&1; cat $O/pass_stamps.txt
el "stamps"
exit 0
