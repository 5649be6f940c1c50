#!/bin/bash
# round 6, GPU call 41: the granule pick-up takes four granules per step (two 32-byte loads, one compare: take4) against the library of
# f7cc14a, alternating; parity tests first (the k_pass of f7cc14a with its dead branch removed is in BOTH libraries).

R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call41; mkdir -p $O
export TMPDIR=/tmp
L=$R/fast_lio_amd/lib
cd $R
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_z_fullsize.py tests/test_gpu_eight_ranks.py tests/test_golden.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee $O/gpu_tests.txt
el "tests"
for rep in 1 2 3; do
  for ext in 1 0; do
    for v in old:$L/libfastlio_hip_old.so new:$L/libfastlio_hip.so; do
      IFS=: read name lib <<< "$v"
      FLH_LIB=$lib timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --extrinsic-est $ext > $O/bench300_ext${ext}_${name}_$rep.json 2> $O/bench300_ext${ext}_${name}_$rep.err
      echo "ext $ext $name rep $rep: $(python tools/bench_line.py $O/bench300_ext${ext}_${name}_$rep.json)"
    done
  done
done
el "done"
exit 0
