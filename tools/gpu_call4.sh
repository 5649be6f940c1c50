#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/c4; mkdir -p $O; cd $R
export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
line() { python $R/tools/bench_line.py "$1"; }
timeout 900 python -m pytest tests -q -m gpu -x -rxXs 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -25 | tee $O/gpu_tests.txt
el "gpu suite done"
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-scans 0 --no-extra-legs > $O/b20.json 2>$O/b20.err; echo "20 steps rc=$?"; line $O/b20.json
for fs in 0 1; do
  timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --first-stage $fs > $O/b300_fs$fs.json 2>$O/b300_fs$fs.err; echo "300 steps fs=$fs rc=$?"; line $O/b300_fs$fs.json
done
cd /tmp
for fs in 0 1; do
  rm -rf /tmp/kt$fs; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$fs -o t -- python $R/bench.py --steps 150 --warmup 20 --cpu-scans 0 --no-extra-legs --in-process --first-stage $fs > /dev/null 2>$O/kt$fs.err
  echo "kernel trace fs=$fs rc=$?"; f=$(find /tmp/kt$fs -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_fs$fs.csv && python $R/tools/kstats.py $f 12
done
el "all done"
