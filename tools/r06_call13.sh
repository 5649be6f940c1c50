#!/bin/bash
# round 6, GPU call 13: the mailbox of the pre-launched no-search pass read in ONE PCIe round trip (sequence word + checksum in both
# 64-byte lines of the host box, FLH_MAIL1) against the product (poll the word, then read the state), alternating.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call13; mkdir -p $O
export TMPDIR=/tmp
L=$R/fast_lio_amd/lib
cd $R
FLH_LIB=$L/libfastlio_hip_mail1.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_zz_timing.py tests/test_gpu_map.py -q -m gpu > $O/mail1_tests.txt 2>&1; grep -E "passed|failed|error" $O/mail1_tests.txt | tail -3
for rep in 1 2 3; do
  for v in head:$L/libfastlio_hip.so mail1:$L/libfastlio_hip_mail1.so; do
    IFS=: read name lib <<< "$v"
    FLH_LIB=$lib timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs > $O/bench300_${name}_$rep.json 2> $O/bench300_${name}_$rep.err
    echo "300 steps $name rep $rep: $(python tools/bench_line.py $O/bench300_${name}_$rep.json)"
    python - $O/bench300_${name}_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("    prelaunched", d.get("prelaunched_nosearch_passes"))
PY
  done
done
for rep in 1 2; do
  for v in head:$L/libfastlio_hip.so mail1:$L/libfastlio_hip_mail1.so; do
    IFS=: read name lib <<< "$v"
    FLH_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 5 --cpu-scans 0 --no-extra-legs > $O/bench20_${name}_$rep.json 2> $O/bench20_${name}_$rep.err
    echo "driver cmd $name rep $rep: $(python tools/bench_line.py $O/bench20_${name}_$rep.json)"
  done
done
exit 0
