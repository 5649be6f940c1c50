#!/bin/bash
# round 6, GPU call 35 (diagnostic): the slow regions of the pipelined leg (call 34: 4 of 30 regions 13-17 % low with normal pass
# times).  bench.py --diag-staging: per region the staging thread's enqueue time and its wait for the H2D copy, the update thread's
# wait in flh_scan_activate.  Product against -DFLH_STAGE_SPIN (the copy's end polled, not slept for), ten 300-step regions per run.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call35; mkdir -p $O
export TMPDIR=/tmp
L=$R/fast_lio_amd/lib
cd $R
nproc; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket|NUMA node\(s\)" 
for rep in 1 2 3; do
  for v in head:$L/libfastlio_hip.so spin:$L/libfastlio_hip_spin.so; do
    IFS=: read name lib <<< "$v"
    FLH_LIB=$lib timeout 300 python bench.py --steps 300 --warmup 30 --repeats 9 --cpu-scans 0 --no-extra-legs --diag-staging > $O/diag_${name}_$rep.json 2> $O/diag_${name}_$rep.err
    python - $O/diag_${name}_$rep.json $name $rep <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[3], "value", d["value"], "resident", d.get("device_resident_scans_per_s"))
for r in d.get("staging_diag", []):
    print("   ", r)
PY
  done
done
exit 0
