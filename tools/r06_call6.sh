#!/bin/bash
# round 6, GPU call 6: staging v4 + k_add_resolve in one walk + the voxel table emptied by the compaction kernel (consecutive tiles, 32-bit keys, windows of eight tiles side by side in LDS, wide windows searched
# where they lie): bits, the kernels alone, then alternating pairs against the vendor sort.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call6; mkdir -p $O
export TMPDIR=/tmp
cd $R
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
timeout 900 python -m pytest tests/test_gpu_staging.py tests/test_gpu_map.py tests/test_gpu_z_fullsize.py -q -m gpu -x > $O/new_tests.txt 2>&1; grep -E "passed|failed|error|Error|assert" $O/new_tests.txt | head -20
el "staging tests"
for n in 100000 200000; do
  cd /tmp; rm -rf /tmp/sp
  PYTHONPATH=$R timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o t -- python $R/tools/stage_probe.py --stage-sort 1 --n $n > $O/stage_probe_ss1_$n.txt 2>&1
  f=$(find /tmp/sp -name '*kernel_stats.csv' | head -1)
  echo "== staging alone: N=$n"; grep "us per" $O/stage_probe_ss1_$n.txt; [ -n "$f" ] && cp $f $O/stage_alone_ss1_$n.csv && python $R/tools/kstats.py $f 30 | grep "k_stage" | head -4
  cd $R
done
el "staging alone"
for rep in 1 2 3; do
  for ss in 0 1; do
    timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --stage-sort $ss > $O/bench300_ss${ss}_$rep.json 2> $O/bench300_ss${ss}_$rep.err
    echo "300 steps stage_sort=$ss rep $rep: $(python tools/bench_line.py $O/bench300_ss${ss}_$rep.json)"
  done
done
el "A/B 300"
for rep in 1 2 3; do
  for ss in 0 1; do
    timeout 300 python bench.py --steps 20 --warmup 5 --cpu-scans 0 --stage-sort $ss > $O/bench20_ss${ss}_$rep.json 2> $O/bench20_ss${ss}_$rep.err
    echo "driver cmd stage_sort=$ss rep $rep: $(python tools/bench_line.py $O/bench20_ss${ss}_$rep.json)"
  done
done
el "A/B 20"
for cfg in 3 5; do
  for ss in 0 1; do
    timeout 300 python bench.py --config $cfg --steps 60 --warmup 6 --scans 32 --cpu-scans 0 --no-extra-legs --stage-sort $ss > $O/bench_config${cfg}_ss$ss.json 2> $O/bench_config${cfg}_ss$ss.err
    echo "config $cfg stage_sort=$ss: $(python tools/bench_line.py $O/bench_config${cfg}_ss$ss.json)"
  done
done
timeout 900 python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; grep -E "passed|failed|error" $O/gpu_tests.txt | tail -5
el "done"
exit 0
