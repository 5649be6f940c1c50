#!/bin/bash
# GPU box, round 5, call 4 (diagnostic): the kernel timeline of the pipelined loop with the pre-launched no-search pass on and off
# (what runs beside k_pass, what the gaps are), the same A/B without the staging sort, and the test call 3 left red.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05_call4; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
cd $R
timeout 600 python -m pytest -q -m gpu tests/test_gpu_z_fullsize.py::test_config2_full_update_against_oracle 2>&1 | tail -5 | cut -c1-200
el "config-2 full-size test"
P="--steps 80 --warmup 20 --cpu-scans 0 --no-extra-legs --in-process --repeats 0"
for v in on off; do
  cd /tmp; rm -rf /tmp/tl_$v
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$v -o t -- python $R/bench.py $P --prelaunch $([ $v = on ] && echo 1 || echo 0) > $O/tl_$v.json 2> $O/tl_$v.err
  f=$(find /tmp/tl_$v -name '*kernel_trace.csv' | head -1)
  if [ -n "$f" ]; then cp $f $O/kernel_trace_prelaunch_$v.csv; head -1 $f | cut -c1-300; python $R/tools/timeline.py $f "prelaunch $v:" | tee $O/timeline_prelaunch_$v.txt; else echo "no trace ($v)"; tail -3 $O/tl_$v.err; fi
  cd $R
done
el "timelines"
B="--steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs"
for rep in 1 2; do
  for v in sort1_on sort1_off sort0_on sort0_off; do
    case $v in
      sort1_on) A="";; sort1_off) A="--prelaunch 0";; sort0_on) A="--sort 0";; sort0_off) A="--sort 0 --prelaunch 0";;
    esac
    timeout 300 python bench.py $B $A > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err; echo "$v $rep rc=$?"; python tools/bench_line.py $O/bench_${v}_$rep.json
  done
done
el "sort / prelaunch A/B"
exit 0
