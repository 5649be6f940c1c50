#!/bin/bash
# GPU box, round 4, call 1: the GPU suite with the one-launch pass, the driver's command with it and with the three-launch pass
# (same box: A/B), the kernel trace of the headline config, and the register-bound variants of k_pass.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_call1; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
cd $R
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 | tee $O/gpu_tests.txt
el "gpu suite"
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/driver.err; echo "driver command rc=$?"; tail -3 $O/driver.err; python tools/bench_line.py $O/bench_driver_cmd.json
el "driver command"
for pk in 1 0; do
  timeout 300 python bench.py --steps 300 --warmup 30 --pass-kernel $pk --cpu-scans 0 --no-extra-legs > $O/bench_pk$pk.json 2> $O/bench_pk$pk.err; echo "pass-kernel $pk rc=$?"; python tools/bench_line.py $O/bench_pk$pk.json
done
el "A/B"
for v in w6 w5u8; do
  if [ -f $R/fast_lio_amd/lib/libfastlio_hip_$v.so ]; then
    FLH_LIB=$R/fast_lio_amd/lib/libfastlio_hip_$v.so timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs > $O/bench_$v.json 2> $O/bench_$v.err; echo "variant $v rc=$?"; python tools/bench_line.py $O/bench_$v.json
  fi
done
el "variants"
cd /tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python $R/bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --in-process > /dev/null 2>$O/kt.err
f=$(find /tmp/kt -name '*kernel_stats.csv' 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_config2.csv && python $R/tools/kstats.py $f 12
el "kernel trace"
cd /tmp; rm -rf /tmp/kt0
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt0 -o t -- python $R/bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --in-process --pass-kernel 0 > /dev/null 2>$O/kt0.err
f=$(find /tmp/kt0 -name '*kernel_stats.csv' 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_config2_three_launch.csv && python $R/tools/kstats.py $f 8
el "kernel trace (three-launch)"
exit 0
