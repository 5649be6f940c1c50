#!/bin/bash
# GPU box, round 4, call 5: the two tests that failed in call 4 (full output), which waves of k_pass are the slow ones (stamps with
# hardware ids and candidate counts), round 3's tree on the same box (did the first stage get slower, or is it the box?), issue and
# traffic counters of k_pass.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_call5; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
cd $R
timeout 600 python -m pytest -q -m gpu --tb=long tests/test_gpu_parity.py::test_rccl_allreduce_path_single_rank tests/test_gpu_peers.py > $O/failed_tests.txt 2>&1; tail -5 $O/failed_tests.txt
el "tests"
FLH_LIB=$R/fast_lio_amd/lib/libfastlio_hip_stamps.so timeout 300 python tools/pass_stamps.py > $O/pass_stamps.txt 2>&1; tail -75 $O/pass_stamps.txt
el "stamps"
if [ -d $R/.r3tree ]; then
  cd /tmp; rm -rf /tmp/kt3
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -o t -- python $R/.r3tree/bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --in-process --cache-dir $R/.bench_cache > $O/bench_r3tree.json 2>$O/kt3.err
  f=$(find /tmp/kt3 -name '*kernel_stats.csv' 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_config2_r3tree.csv && python $R/tools/kstats.py $f 6
  python $R/tools/bench_line.py $O/bench_r3tree.json
  el "round-3 tree"
fi
cd /tmp
rm -rf /tmp/sq2; timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d /tmp/sq2 -o c -- python $R/bench.py --config 2 --steps 40 --warmup 5 --cpu-scans 0 --no-extra-legs --in-process > /dev/null 2>$O/sq2.err
f=$(find /tmp/sq2 -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py $O/pmc_sq_config2.csv $f | grep "k_pass\|k_search\|k_fit"
el "SQ counters"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pm_$c -o c -- python $R/bench.py --config 2 --steps 40 --warmup 5 --cpu-scans 0 --no-extra-legs --in-process > /dev/null 2>$O/pm_$c.err
done
fa=$(find /tmp/pm_FETCH_SIZE -name '*counter_collection.csv' | head -1); fb=$(find /tmp/pm_WRITE_SIZE -name '*counter_collection.csv' | head -1)
[ -n "$fa" ] && [ -n "$fb" ] && python $R/tools/pmc_summary.py $O/pmc_summary_config2.csv $fa $fb | grep "k_pass\|k_search\|k_fit"
el "traffic counters"
exit 0
