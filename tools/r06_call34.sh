#!/bin/bash
# round 6, GPU call 34: the granule pick-up requests the lines of the next two groups ahead (flh_api.cpp: collect_granules,
# prefetch_group) against the library of f34d33d, alternating on one box: config 2 at 300 steps with extrinsic_est_en 1 and 0.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call34; mkdir -p $O
export TMPDIR=/tmp
L=$R/fast_lio_amd/lib
cd $R
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee $O/gpu_tests.txt
el "parity tests"
for rep in 1 2 3; do
  for ext in 1 0; do
    for v in old:$L/libfastlio_hip_nopf.so new:$L/libfastlio_hip.so; do
      IFS=: read name lib <<< "$v"
      FLH_LIB=$lib timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --extrinsic-est $ext > $O/bench300_ext${ext}_${name}_$rep.json 2> $O/bench300_ext${ext}_${name}_$rep.err
      echo "ext $ext $name rep $rep: $(python tools/bench_line.py $O/bench300_ext${ext}_${name}_$rep.json)"
    done
  done
done
el "done"
exit 0
