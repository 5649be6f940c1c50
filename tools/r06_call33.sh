#!/bin/bash
# round 6, GPU call 33 (diagnostic): where do the extrinsic columns cost?  Kernel times of k_pass / k_fit with extrinsic_est_en 0 and 1
# under rocprofv3 --kernel-trace (--prelaunch 0: the no-search pass launched the usual way, so that its kernel time is the pass's).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call33; mkdir -p $O
export TMPDIR=/tmp
cd $R
for ext in 0 1 0 1; do
  n=$(ls $O | grep -c "kstats_ext${ext}_")
  cd /tmp; rm -rf /tmp/kt
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python $R/bench.py --cpu-scans 0 --no-extra-legs --in-process --prelaunch 0 --extrinsic-est $ext --steps 100 --warmup 10 > $O/line_ext${ext}_$n.json 2>$O/kt_ext${ext}_$n.err
  f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kstats_ext${ext}_$n.csv && echo "== ext $ext" && python $R/tools/kstats.py $f 6
  cd $R
done
exit 0
