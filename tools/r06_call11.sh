#!/bin/bash
# round 6, GPU call 11: why is the driver's command (25 scans, each seen for the first time) 8-10 % below the 300-step line?
# The same 20 timed steps behind warm-ups of different length: 5 (the driver's), 133 (every scan of the 128 seen once before it is
# timed), and 5 with twice as many distinct scans.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call11; mkdir -p $O
export TMPDIR=/tmp
cd $R
for rep in 1 2; do
  for w in 5 133 30; do
    timeout 300 python bench.py --steps 20 --warmup $w --cpu-scans 0 --no-extra-legs > $O/bench20_w${w}_$rep.json 2> $O/bench20_w${w}_$rep.err
    echo "steps 20 warmup $w rep $rep: $(python tools/bench_line.py $O/bench20_w${w}_$rep.json)"
  done
done
exit 0
