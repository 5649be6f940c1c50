#!/bin/bash
# round 6, GPU call 45: does the second staging lane still earn its keep now that the staging is two kernels?  (VERDICT r5 item 1:
# "the second staging lane deleted if one lane now suffices (measure --ring 2 vs --ring 4)").  --ring 2 = one scan staged ahead (one
# staging in flight at a time), 3 (the default) and 4 = two ahead on two lanes.  Alternating, 300 steps, and the driver's command.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call45; mkdir -p $O
export TMPDIR=/tmp
cd $R
for rep in 1 2 3; do
  for ring in 2 3 4; do
    timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --ring $ring > $O/bench300_ring${ring}_$rep.json 2> $O/bench300_ring${ring}_$rep.err
    echo "300 steps ring $ring rep $rep: $(python tools/bench_line.py $O/bench300_ring${ring}_$rep.json)"
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scans 0 --no-extra-legs --ring $ring > $O/bench20_ring${ring}_$rep.json 2> $O/bench20_ring${ring}_$rep.err
    echo "driver cmd ring $ring rep $rep: $(python tools/bench_line.py $O/bench20_ring${ring}_$rep.json)"
  done
done
exit 0
