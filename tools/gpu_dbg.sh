#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/dbg2; mkdir -p $O; cd $R
export TMPDIR=/tmp
B="python bench.py --cpu-scans 0 --no-extra-legs"
$B --steps 20 --warmup 5 > $O/a.json 2>/dev/null; echo -n "steps 20 warmup 5: "; python tools/bench_line.py $O/a.json
$B --steps 20 --warmup 60 > $O/b.json 2>/dev/null; echo -n "steps 20 warmup 60: "; python tools/bench_line.py $O/b.json
$B --steps 20 --warmup 5 --timing-samples 1 > $O/c.json 2>/dev/null; echo -n "steps 20 warmup 5, almost no events: "; python tools/bench_line.py $O/c.json
$B --steps 80 --warmup 5 > $O/d.json 2>/dev/null; echo -n "steps 80 warmup 5: "; python tools/bench_line.py $O/d.json
$B --steps 20 --warmup 5 --scans 25 > $O/e.json 2>/dev/null; echo -n "steps 20 warmup 5, 25 scans: "; python tools/bench_line.py $O/e.json
