#!/bin/bash
# GPU box, round 4, call 6: the two re-toleranced tests, the launch probe (what a pass costs beside its kernel), then the fault
# hunt with the bounds-instrumented library (tools/fault_hunt.sh).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_call6; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
cd $R
timeout 600 python -m pytest -q -m gpu --tb=long tests/test_gpu_parity.py::test_rccl_allreduce_path_single_rank tests/test_gpu_peers.py > $O/failed_tests.txt 2>&1; tail -5 $O/failed_tests.txt
el "tests"
timeout 120 tools/launch_probe > $O/launch_probe.txt 2>&1; cat $O/launch_probe.txt
el "launch probe"
NA=${NA:-264} NB=${NB:-6} NC=${NC:-40} timeout 1000 bash tools/fault_hunt.sh 2>&1 | tail -14
el "fault hunt"
exit 0
