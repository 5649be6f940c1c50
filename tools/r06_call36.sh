#!/bin/bash
# round 6, GPU call 36 (diagnostic): does the box-to-box / region-to-region spread of the pipelined leg come from WHERE the host
# threads run?  The box has two sockets (2 x EPYC 9575F, 2 NUMA nodes); the GPU hangs off one of them.  bench.py under taskset:
# the GPU's own node, the other node, no binding -- the driver's command and 300-step runs with --diag-staging.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call36; mkdir -p $O
export TMPDIR=/tmp
cd $R
for d in /sys/bus/pci/devices/*; do
  [ "$(cat $d/vendor 2>/dev/null)" = "0x1002" ] || continue
  c=$(cat $d/class 2>/dev/null); case $c in 0x0302*|0x0380*|0x0300*|0x1200*) ;; *) continue;; esac
  echo "$(basename $d) class $c numa_node $(cat $d/numa_node) local_cpulist $(cat $d/local_cpulist)"
done | tee $O/gpus.txt
python - <<'PY' | tee $O/visible.txt
import torch
p = torch.cuda.get_device_properties(0)
print("visible device 0:", p.name, "pci", getattr(p, "pci_domain_id", None), getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None))
PY
BDF=$(python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print("%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))
PY
)
NODE=$(cat /sys/bus/pci/devices/$BDF/numa_node); LOCAL=$(cat /sys/bus/pci/devices/$BDF/local_cpulist)
echo "GPU $BDF numa_node $NODE local_cpulist $LOCAL"
OTHER=$(( NODE == 0 ? 1 : 0 )); REMOTE=$(cat /sys/devices/system/node/node$OTHER/cpulist)
echo "remote node $OTHER cpulist $REMOTE"
cat /proc/self/status | grep -i "cpus_allowed_list\|mems_allowed_list"
line() { python - $1 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   value", d["value"], "repeats", (d.get("value_repeats") or {}).get("all"), "resident", d.get("device_resident_scans_per_s"), "search/no-search us", d["ms_search_pass"] * 1e3, d["ms_nosearch_pass"] * 1e3)
for r in d.get("staging_diag", [])[:1]:
    print("   first region:", {k: r[k] for k in ("stage_enq_us", "h2d_wait_us", "act_wait_us", "act_wait_max_us")})
PY
}
for rep in 1 2; do
  for b in none:"" local:"taskset -c $LOCAL" remote:"taskset -c $REMOTE"; do
    name=${b%%:*}; pre=${b#*:}
    $pre timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scans 0 --no-extra-legs > $O/bench20_${name}_$rep.json 2> $O/bench20_${name}_$rep.err
    echo "driver cmd, $name, rep $rep"; line $O/bench20_${name}_$rep.json
    $pre timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs --diag-staging > $O/bench300_${name}_$rep.json 2> $O/bench300_${name}_$rep.err
    echo "300 steps, $name, rep $rep"; line $O/bench300_${name}_$rep.json
  done
done
exit 0
