#!/bin/bash
# round 6, GPU call 3: (a) the staging kernels standing alone (no update beside them): v2 (HEAD), v1 (first version, ballots + global
# binary searches), vendor sort; (b) k_pass variant FLH_V6 (packed (dx, dy), no evaluation-side mask): bits, then alternating pairs.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call3; mkdir -p $O
export TMPDIR=/tmp
L=$R/fast_lio_amd/lib
cd $R
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
for v in v2:$L/libfastlio_hip.so:1 v1:$L/libfastlio_hip_stagev1.so:1 vendor:$L/libfastlio_hip.so:0; do
  IFS=: read name lib ss <<< "$v"
  for n in 100000 200000; do
    cd /tmp; rm -rf /tmp/sp
    FLH_LIB=$lib PYTHONPATH=$R timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o t -- python $R/tools/stage_probe.py --stage-sort $ss --n $n > $O/stage_probe_${name}_$n.txt 2>&1
    f=$(find /tmp/sp -name '*kernel_stats.csv' | head -1)
    echo "== staging alone: $name N=$n"; grep "us per" $O/stage_probe_${name}_$n.txt; [ -n "$f" ] && cp $f $O/stage_alone_${name}_$n.csv && python $R/tools/kstats.py $f 30 | grep -v "k_map\|k_brick\|k_aabb\|k_fill_tomb\|fillBuffer\|copyBuffer" | head -16
    cd $R
  done
done
el "staging alone"
FLH_LIB=$L/libfastlio_hip_v6.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_zz_timing.py tests/test_gpu_z_fullsize.py tests/test_golden.py -q -m gpu > $O/v6_tests.txt 2>&1; grep -E "passed|failed|error" $O/v6_tests.txt | tail -3
el "v6 bits"
for rep in 1 2 3; do
  for v in head:$L/libfastlio_hip.so v6:$L/libfastlio_hip_v6.so; do
    IFS=: read name lib <<< "$v"
    FLH_LIB=$lib timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs > $O/bench300_${name}_$rep.json 2> $O/bench300_${name}_$rep.err
    echo "300 steps $name rep $rep: $(python tools/bench_line.py $O/bench300_${name}_$rep.json)"
  done
done
el "v6 pairs"
cd /tmp; rm -rf /tmp/sq6
FLH_LIB=$L/libfastlio_hip_v6.so timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d /tmp/sq6 -o c -- python $R/bench.py --cpu-scans 0 --no-extra-legs --in-process --prelaunch 0 --steps 40 --warmup 5 > /dev/null 2> $O/sq6.err
f=$(find /tmp/sq6 -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py $O/pmc_sq_v6.csv $f | grep "k_pass\|k_fit"
rm -rf /tmp/sq0
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d /tmp/sq0 -o c -- python $R/bench.py --cpu-scans 0 --no-extra-legs --in-process --prelaunch 0 --steps 40 --warmup 5 > /dev/null 2> $O/sq0.err
f=$(find /tmp/sq0 -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py $O/pmc_sq_head.csv $f | grep "k_pass\|k_fit"
el "done"
exit 0
