#!/bin/bash
# round 6, GPU call 7: staging v4 with branch-free searches: alone and in the pipeline; config 3's kernels after the map-change
# work; what one GPU can say about the 8-way split of BASELINE configs[3] and [4] (tools/exchange_probe.py).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call7; mkdir -p $O
export TMPDIR=/tmp
cd $R
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
timeout 900 python -m pytest tests/test_gpu_staging.py tests/test_abi.py -q -m gpu -x > $O/new_tests.txt 2>&1; grep -E "passed|failed|error|Error|assert" $O/new_tests.txt | head -20
el "staging tests"
for n in 100000 60000; do
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
for rep in 1 2; do
  for ss in 0 1; do
    timeout 300 python bench.py --steps 20 --warmup 5 --cpu-scans 0 --stage-sort $ss > $O/bench20_ss${ss}_$rep.json 2> $O/bench20_ss${ss}_$rep.err
    echo "driver cmd stage_sort=$ss rep $rep: $(python tools/bench_line.py $O/bench20_ss${ss}_$rep.json)"
  done
done
el "A/B 20"
cd /tmp; rm -rf /tmp/kt3
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -o t -- python $R/bench.py --config 3 --steps 100 --warmup 10 --scans 32 --cpu-scans 0 --no-extra-legs --in-process --prelaunch 0 > /dev/null 2> $O/kt3.err
f=$(find /tmp/kt3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_config3.csv && python $R/tools/kstats.py $f 20
cd $R
for ss in 0 1 0 1; do
  timeout 300 python bench.py --config 3 --steps 100 --warmup 10 --scans 32 --cpu-scans 0 --no-extra-legs --stage-sort $ss > $O/bench_config3_ss${ss}.json 2> $O/bench_config3_ss$ss.err
  echo "config 3 stage_sort=$ss: $(python tools/bench_line.py $O/bench_config3_ss${ss}.json)"
done
el "config 3"
timeout 500 python tools/exchange_probe.py --config 4 --only-shares > $O/exchange_probe_config4.txt 2>&1; cat $O/exchange_probe_config4.txt | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
el "exchange probe config 4"
timeout 700 python tools/exchange_probe.py --config 5 > $O/exchange_probe_config5.txt 2>&1; cat $O/exchange_probe_config5.txt | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
el "exchange probe config 5"
timeout 300 python tools/exchange_probe.py --config 2 > $O/exchange_probe_config2.txt 2>&1; cat $O/exchange_probe_config2.txt | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
el "done"
exit 0
