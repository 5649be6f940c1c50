#!/bin/bash
# round 6, GPU call 12: k_pass with the fit wave at raised priority (s_setprio 1 / 3) against the product, alternating.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call12; mkdir -p $O
export TMPDIR=/tmp
L=$R/fast_lio_amd/lib
cd $R
for rep in 1 2 3; do
  for v in head:$L/libfastlio_hip.so prio3:$L/libfastlio_hip_prio3.so prio1:$L/libfastlio_hip_prio1.so; do
    IFS=: read name lib <<< "$v"
    FLH_LIB=$lib timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs > $O/bench300_${name}_$rep.json 2> $O/bench300_${name}_$rep.err
    echo "300 steps $name rep $rep: $(python tools/bench_line.py $O/bench300_${name}_$rep.json)"
  done
done
exit 0
