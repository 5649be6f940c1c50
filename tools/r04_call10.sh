#!/bin/bash
# GPU box, round 4, call 10: the exchange probe (what the multi-GPU split can gain, measured on one GPU), then the second batch of
# the fault hunt with the bounds-instrumented build of the final sources (8 fresh scans per run instead of 32: 4x the runs per minute).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_call10; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
cd $R
timeout 240 python tools/exchange_probe.py > $O/exchange_probe.txt 2>&1; cat $O/exchange_probe.txt
el "exchange probe"
NA=${NA:-400} NB=${NB:-3} NC=${NC:-16} SCANS=8 BUDGET=${BUDGET:-300} timeout 520 bash tools/fault_hunt.sh 2>&1 | tail -12
el "fault hunt"
exit 0
