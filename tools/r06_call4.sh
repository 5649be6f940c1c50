#!/bin/bash
# round 6, GPU call 4: staging v3 (round-robin tiles, ballots, composites, interleaved LDS searches) + map_incremental without the
# gather kernel and the library scan: bits first, then the kernels alone, then same-box A/B.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call4; mkdir -p $O
export TMPDIR=/tmp
cd $R
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
timeout 600 python -m pytest tests/test_gpu_staging.py tests/test_gpu_map.py -q -m gpu -x > $O/new_tests.txt 2>&1; grep -E "passed|failed|error|Error|assert" $O/new_tests.txt | head -20
el "staging + map tests"
timeout 900 python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; grep -E "passed|failed|error" $O/gpu_tests.txt | tail -5
el "gpu suite"
for ss in 1 0; do
  for n in 100000 200000; do
    cd /tmp; rm -rf /tmp/sp
    PYTHONPATH=$R timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o t -- python $R/tools/stage_probe.py --stage-sort $ss --n $n > $O/stage_probe_ss${ss}_$n.txt 2>&1
    f=$(find /tmp/sp -name '*kernel_stats.csv' | head -1)
    echo "== staging alone: stage_sort=$ss N=$n"; grep "us per" $O/stage_probe_ss${ss}_$n.txt; [ -n "$f" ] && cp $f $O/stage_alone_ss${ss}_$n.csv && python $R/tools/kstats.py $f 30 | grep "k_stage\|k_merge_config\|k_sort_config\|k_scan_restride\|k_scan_gather\|rocprim" | head -6
    cd $R
  done
done
el "staging alone"
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
cd /tmp
for cfg in 2 3; do
  rm -rf /tmp/kt$cfg
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$cfg -o t -- python $R/bench.py --config $cfg $([ $cfg = 3 ] && echo "--steps 100 --warmup 10 --scans 32") --cpu-scans 0 --no-extra-legs --in-process --prelaunch 0 > /dev/null 2> $O/kt$cfg.err
  f=$(find /tmp/kt$cfg -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_config$cfg.csv && python $R/tools/kstats.py $f 18
done
cd $R
for rep in 1 2; do
  timeout 300 python bench.py --config 3 --steps 100 --warmup 10 --scans 32 --cpu-scans 0 --no-extra-legs > $O/bench_config3_$rep.json 2> $O/bench_config3_$rep.err
  echo "config 3 rep $rep: $(python tools/bench_line.py $O/bench_config3_$rep.json)"
done
for cfg in 4 5; do
  timeout 300 python bench.py --config $cfg --steps 60 --warmup 6 --scans 32 --cpu-scans 0 --no-extra-legs > $O/bench_config$cfg.json 2> $O/bench_config$cfg.err
  echo "config $cfg: $(python tools/bench_line.py $O/bench_config$cfg.json)"
done
el "done"
exit 0
