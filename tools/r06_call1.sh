#!/bin/bash
# round 6, GPU call 1: the library's own staging kernels (flh_stage.hip) -- parity first, then same-box A/B against the vendor sort.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call1; mkdir -p $O
export TMPDIR=/tmp
cd $R
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
timeout 600 python -m pytest tests/test_gpu_staging.py tests/test_golden.py -q -m gpu -x 2>&1 | tail -15 | tee $O/staging_tests.txt
el "staging tests"
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee $O/gpu_tests.txt
el "gpu suite"
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
for ring in 2 3 4; do
  timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --stage-sort 1 --ring $ring > $O/bench300_ring$ring.json 2> $O/bench300_ring$ring.err
  echo "ring $ring: $(python tools/bench_line.py $O/bench300_ring$ring.json)"
done
el "rings"
cd /tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python $R/bench.py --cpu-scans 0 --no-extra-legs --in-process --prelaunch 0 > /dev/null 2> $O/kt.err
f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_config2.csv && python $R/tools/kstats.py $f 14
cd $R
for cfg in 3 5; do
  timeout 300 python bench.py --config $cfg --steps 60 --warmup 6 --scans 32 --cpu-scans 0 --no-extra-legs > $O/bench_config$cfg.json 2> $O/bench_config$cfg.err
  echo "config $cfg: $(python tools/bench_line.py $O/bench_config$cfg.json)"
done
el "done"
exit 0
