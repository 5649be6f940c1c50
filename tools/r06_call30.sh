#!/bin/bash
# round 6, GPU call 30: k_pass with chunks of 4 / 16 consecutive units on ONE XCD (FLH_XCD_CHUNK: neighbouring units share search
# cells; dealt round-robin they fetch them into eight L2s) against the product, alternating; parity of the variants first.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call30; mkdir -p $O
export TMPDIR=/tmp
L=$R/fast_lio_amd/lib
cd $R
for v in xcd4 xcd16; do
  FLH_LIB=$L/libfastlio_hip_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -q -m gpu -k "not reference_operation_sequence" > $O/${v}_tests.txt 2>&1; echo "$v: $(grep -E 'passed|failed' $O/${v}_tests.txt | tail -1)"
done
for rep in 1 2 3; do
  for v in head:$L/libfastlio_hip.so xcd4:$L/libfastlio_hip_xcd4.so xcd16:$L/libfastlio_hip_xcd16.so; do
    IFS=: read name lib <<< "$v"
    FLH_LIB=$lib timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs > $O/bench300_${name}_$rep.json 2> $O/bench300_${name}_$rep.err
    echo "300 steps $name rep $rep: $(python tools/bench_line.py $O/bench300_${name}_$rep.json)"
  done
done
for v in head:$L/libfastlio_hip.so xcd4:$L/libfastlio_hip_xcd4.so xcd16:$L/libfastlio_hip_xcd16.so; do
  IFS=: read name lib <<< "$v"
  cd /tmp; rm -rf /tmp/tcc
  FLH_LIB=$lib timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/tcc -o c -- python $R/bench.py --cpu-scans 0 --no-extra-legs --in-process --prelaunch 0 --steps 40 --warmup 5 > /dev/null 2>$O/tcc_$name.err
  f=$(find /tmp/tcc -name '*counter_collection.csv' | head -1); [ -n "$f" ] && echo "== L2 $name" && python $R/tools/pmc_summary.py $O/pmc_l2_$name.csv $f | grep "k_pass"
  cd $R
done
exit 0
