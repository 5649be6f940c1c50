#!/bin/bash
# round 6, GPU call 32: the filter's step from ONE right-hand side, K_x formed only in the pass that ends the update
# (include/fastlio_amd/esekfom.hpp: info_step) against the library of 217c485 (K_h, K_x and 23 right-hand sides in every pass),
# alternating on one box: the driver's command, config 2 at 300 steps with extrinsic_est_en 0 and 1, config 3; the host-side GPU
# tests first.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call32; mkdir -p $O
export TMPDIR=/tmp
L=$R/fast_lio_amd/lib
cd $R
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
python tools/src_hash.py
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_z_fullsize.py -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee $O/gpu_tests.txt
el "parity + full-size tests"
for rep in 1 2 3; do
  for v in old:$L/libfastlio_hip_oldalg.so new:$L/libfastlio_hip.so; do
    IFS=: read name lib <<< "$v"
    FLH_LIB=$lib timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scans 0 --no-extra-legs > $O/bench20_${name}_$rep.json 2> $O/bench20_${name}_$rep.err
    echo "driver cmd $name rep $rep: $(python tools/bench_line.py $O/bench20_${name}_$rep.json)"
    python - $O/bench20_${name}_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("    repeats", (d.get("value_repeats") or {}).get("all"))
PY
  done
done
el "driver's command"
for rep in 1 2; do
  for ext in 0 1; do
    for v in old:$L/libfastlio_hip_oldalg.so new:$L/libfastlio_hip.so; do
      IFS=: read name lib <<< "$v"
      FLH_LIB=$lib timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --extrinsic-est $ext > $O/bench300_ext${ext}_${name}_$rep.json 2> $O/bench300_ext${ext}_${name}_$rep.err
      echo "config 2, 300 steps, ext $ext $name rep $rep: $(python tools/bench_line.py $O/bench300_ext${ext}_${name}_$rep.json)"
    done
  done
done
el "config 2"
for rep in 1; do
  for v in old:$L/libfastlio_hip_oldalg.so new:$L/libfastlio_hip.so; do
    IFS=: read name lib <<< "$v"
    FLH_LIB=$lib timeout 300 python bench.py --config 3 --steps 100 --warmup 10 --scans 32 --cpu-scans 0 --no-extra-legs > $O/bench_config3_${name}_$rep.json 2> $O/bench_config3_${name}_$rep.err
    echo "config 3 $name rep $rep: $(python tools/bench_line.py $O/bench_config3_${name}_$rep.json)"
  done
done
el "done"
exit 0
