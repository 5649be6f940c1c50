#!/bin/bash
# GPU box, round 4, call 12: k_pass's units dispatched in the order the staging leaves (sparsest units first, k_unit_order) against
# the plain reversed Morton order (-DFLH_PASS_REVERSED build = the tree of the evidence call), same box; stamps by unit.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_call12; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
cd $R
timeout 400 python -m pytest -q -m gpu tests/test_gpu_zz_timing.py tests/test_gpu_parity.py 2>&1 | tail -30 > $O/gpu_tests.txt; tail -6 $O/gpu_tests.txt | cut -c1-200
el "tests"
for v in order rev order rev; do
  L=$R/fast_lio_amd/lib/libfastlio_hip.so; [ $v = rev ] && L=$R/fast_lio_amd/lib/libfastlio_hip_rev.so
  FLH_LIB=$L timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --timing-samples 64 > $O/bench_$v.json 2> $O/bench_$v.err; echo "$v rc=$?"; python tools/bench_line.py $O/bench_$v.json
done
el "unit order A/B"
FLH_LIB=$R/fast_lio_amd/lib/libfastlio_hip_stamps.so timeout 200 python tools/pass_stamps.py > $O/pass_stamps.txt 2>&1; grep -A 24 "deciles of the DISPATCH" $O/pass_stamps.txt | head -60; grep "last stamp" $O/pass_stamps.txt
el "stamps"
exit 0
