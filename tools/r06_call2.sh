#!/bin/bash
# round 6, GPU call 2: staging kernels v2 (LDS lane-mask ranking, samples + windows in LDS) -- parity with full logs, the 8-rank
# tests, the gate-kernel probe, then same-box A/B against the vendor sort and a kernel trace.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call2; mkdir -p $O
export TMPDIR=/tmp
cd $R
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
timeout 600 python -m pytest tests/test_gpu_staging.py tests/test_golden.py tests/test_gpu_eight_ranks.py -q -m gpu -x > $O/new_tests.txt 2>&1; grep -E "passed|failed|error|Error|assert" $O/new_tests.txt | head -20
el "new tests"
timeout 900 python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; grep -E "passed|failed|error" $O/gpu_tests.txt | tail -5
el "gpu suite"
timeout 120 tools/mailbox_probe > $O/mailbox_probe.txt 2>&1; cat $O/mailbox_probe.txt
el "mailbox probe"
for rep in 1 2; do
  for ss in 0 1; do
    timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --stage-sort $ss > $O/bench300_ss${ss}_$rep.json 2> $O/bench300_ss${ss}_$rep.err
    echo "300 steps stage_sort=$ss rep $rep: $(python tools/bench_line.py $O/bench300_ss${ss}_$rep.json)"
  done
done
el "A/B 300"
for rep in 1 2; do
  for ss in 0 1; do
    timeout 300 python bench.py --steps 20 --warmup 5 --cpu-scans 0 --stage-sort $ss > $O/bench20_ss${ss}_$rep.json 2> $O/bench20_ss${ss}_$rep.err
    echo "driver cmd stage_sort=$ss rep $rep: $(python tools/bench_line.py $O/bench20_ss${ss}_$rep.json)"
  done
done
el "A/B 20"
for ring in 2 4; do
  timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --stage-sort 1 --ring $ring > $O/bench300_ring$ring.json 2> $O/bench300_ring$ring.err
  echo "ring $ring: $(python tools/bench_line.py $O/bench300_ring$ring.json)"
done
el "rings"
cd /tmp
for cfg in 2 5; do
  rm -rf /tmp/kt$cfg
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$cfg -o t -- python $R/bench.py --config $cfg $([ $cfg = 5 ] && echo "--steps 60 --warmup 6 --scans 32") --cpu-scans 0 --no-extra-legs --in-process --prelaunch 0 > /dev/null 2> $O/kt$cfg.err
  f=$(find /tmp/kt$cfg -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_config$cfg.csv && python $R/tools/kstats.py $f 6
done
cd $R
for cfg in 3 5; do
  timeout 300 python bench.py --config $cfg --steps 60 --warmup 6 --scans 32 --cpu-scans 0 --no-extra-legs > $O/bench_config$cfg.json 2> $O/bench_config$cfg.err
  echo "config $cfg: $(python tools/bench_line.py $O/bench_config$cfg.json)"
done
el "done"
exit 0
