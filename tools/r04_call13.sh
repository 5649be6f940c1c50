#!/bin/bash
# GPU box, round 4, call 13: s_setprio(3) for what follows a workgroup's phase A (variant build prio3) against the product, same box;
# the bit-equality tests on the final product library.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_call13; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
cd $R
timeout 300 python -m pytest -q -m gpu tests/test_gpu_zz_timing.py tests/test_gpu_parity.py 2>&1 | tail -4 | tee $O/gpu_tests.txt | cut -c1-200
el "tests"
for v in base prio3 base prio3; do
  L=$R/fast_lio_amd/lib/libfastlio_hip.so; [ $v = prio3 ] && L=$R/fast_lio_amd/lib/libfastlio_hip_prio3.so
  FLH_LIB=$L timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --timing-samples 64 > $O/bench_$v.json 2> $O/bench_$v.err; echo "$v rc=$?"; python tools/bench_line.py $O/bench_$v.json
done
el "s_setprio A/B"
exit 0
