#!/bin/bash
# GPU box, round 4, call 2: the whole GPU suite (no -x), then the fault hunt with the bounds-instrumented library.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_call2; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -40 | tee $O/gpu_tests.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/driver.err; echo "driver command rc=$?"; python tools/bench_line.py $O/bench_driver_cmd.json
NA=${NA:-240} NB=${NB:-6} NC=${NC:-24} timeout 2400 bash tools/fault_hunt.sh 2>&1 | tail -30
exit 0
