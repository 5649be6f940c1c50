#!/bin/bash
# round 6, GPU call 31: FLH_XCD_CHUNK = 16 / 32 / 64 against the product: k_pass time and value (configs 2, 4, 5), FETCH_SIZE /
# WRITE_SIZE / L2 hits of a searching pass (config 2).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call31; mkdir -p $O
export TMPDIR=/tmp
L=$R/fast_lio_amd/lib
cd $R
VARS="head:$L/libfastlio_hip.so xcd16:$L/libfastlio_hip_xcd16.so xcd32:$L/libfastlio_hip_xcd32.so xcd64:$L/libfastlio_hip_xcd64.so"
for v in xcd32 xcd64; do
  FLH_LIB=$L/libfastlio_hip_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -q -m gpu -k "not reference_operation_sequence" > $O/${v}_tests.txt 2>&1; echo "$v: $(grep -E 'passed|failed' $O/${v}_tests.txt | tail -1)"
done
for rep in 1 2; do
  for v in $VARS; do
    IFS=: read name lib <<< "$v"
    FLH_LIB=$lib timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs > $O/bench300_${name}_$rep.json 2> $O/bench300_${name}_$rep.err
    echo "config 2 $name rep $rep: $(python tools/bench_line.py $O/bench300_${name}_$rep.json)"
  done
done
for cfg in 4 5 3; do
  for v in $VARS; do
    IFS=: read name lib <<< "$v"
    FLH_LIB=$lib timeout 300 python bench.py --config $cfg --steps 60 --warmup 6 --scans 32 --cpu-scans 0 --no-extra-legs > $O/bench_config${cfg}_${name}.json 2> $O/bench_config${cfg}_${name}.err
    echo "config $cfg $name: $(python tools/bench_line.py $O/bench_config${cfg}_${name}.json)"
  done
done
for v in $VARS; do
  IFS=: read name lib <<< "$v"
  for c in FETCH_SIZE WRITE_SIZE; do
    cd /tmp; rm -rf /tmp/pm
    FLH_LIB=$lib timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pm -o c -- python $R/bench.py --cpu-scans 0 --no-extra-legs --in-process --prelaunch 0 --steps 40 --warmup 5 > /dev/null 2>$O/pm_${name}_$c.err
    f=$(find /tmp/pm -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/pmc_${name}_$c.csv
    cd $R
  done
  python tools/pmc_summary.py $O/pmc_summary_$name.csv $O/pmc_${name}_FETCH_SIZE.csv $O/pmc_${name}_WRITE_SIZE.csv | grep "k_pass" | sed "s/^/$name /"
done
exit 0
