#!/bin/bash
# GPU box, round 5, call 5: the pre-launched SEARCHING pass (flh_config.prelaunch = 2) -- parity, then alternating A/B against
# prelaunch = 1 (the default) and 0; plus the full-size config-2 test with the extrinsic columns.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05_call5; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
cd $R
timeout 900 python -m pytest -q -m gpu tests/test_gpu_parity.py tests/test_gpu_zz_timing.py "tests/test_gpu_z_fullsize.py::test_config2_full_update_against_oracle" -s 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" > $O/gpu_tests_full.txt
tail -100 $O/gpu_tests_full.txt > $O/gpu_tests.txt
grep -E "passed|failed|^FAILED|^ERROR|^E  " $O/gpu_tests_full.txt | cut -c1-300 | tail -30
el "parity"
timeout 300 python tools/prelaunch_check.py --steps 200 > $O/prelaunch_check.txt 2>&1; echo "prelaunch check rc=$?"; tail -10 $O/prelaunch_check.txt
el "prelaunch A/B in one process (resident scans)"
B="--steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs"
for rep in 1 2 3; do
  for v in 1 2 0; do
    timeout 300 python bench.py $B --prelaunch $v > $O/bench_prelaunch${v}_$rep.json 2> $O/bench_prelaunch${v}_$rep.err; echo "prelaunch $v rep $rep rc=$?"; python tools/bench_line.py $O/bench_prelaunch${v}_$rep.json
  done
done
el "bench A/B"
for v in 1 2; do
  timeout 400 python bench.py --steps 20 --warmup 5 --cpu-scans 0 --no-extra-legs --prelaunch $v > $O/bench_driver_cmd_prelaunch$v.json 2> $O/bench_driver_cmd_prelaunch$v.err; echo "driver command, prelaunch $v rc=$?"; python tools/bench_line.py $O/bench_driver_cmd_prelaunch$v.json
done
el "driver command"
exit 0
