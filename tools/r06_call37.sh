#!/bin/bash
# round 6, GPU call 37: how narrowly should the process be bound?  none / the GPU's NUMA node / the GPU's share of the node's cores
# (node cores divided among the node's GPUs by PCI order: 16 physical cores = 2 CCDs here) with and without the SMT siblings / one CCD.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06_call37; mkdir -p $O
export TMPDIR=/tmp
cd $R
python - > $O/sets.txt <<'PY'
import glob, os, torch
p = torch.cuda.get_device_properties(0)
bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
def parse(s):
    out = []
    for part in s.strip().split(","):
        if "-" in part:
            a, b = part.split("-"); out += list(range(int(a), int(b) + 1))
        elif part: out.append(int(part))
    return out
local = parse(open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read())
gpus = []
for d in sorted(glob.glob("/sys/bus/pci/devices/*")):
    try:
        if open(d + "/vendor").read().strip() == "0x1002" and open(d + "/class").read().strip().startswith(("0x1200", "0x0302", "0x0380")) and int(open(d + "/numa_node").read()) == node:
            gpus.append(os.path.basename(d))
    except OSError: pass
idx = gpus.index(bdf)
# physical cores of the node = CPUs that are the first of their thread_siblings_list
phys = [c for c in local if parse(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read())[0] == c]
per = len(phys) // max(len(gpus), 1)
share = phys[idx * per:(idx + 1) * per]
sibs = sorted(set(sum((parse(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read()) for c in share), [])))
l3 = parse(open(f"/sys/devices/system/cpu/cpu{share[0]}/cache/index3/shared_cpu_list").read())
ccd = [c for c in l3 if c in phys]
f = lambda v: ",".join(map(str, v))
print("node", f(local)); print("share", f(share)); print("sharesmt", f(sibs)); print("ccd", f(ccd))
import sys
print(f"# GPU {bdf} node {node} index {idx} of {len(gpus)} on the node; {len(phys)} physical cores; L3 group of cpu {share[0]}: {l3}", file=sys.stderr)
PY
cat $O/sets.txt
line() { python - $1 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   value", d["value"], "repeats", (d.get("value_repeats") or {}).get("all"), "resident", d.get("device_resident_scans_per_s"), "search/no-search us", round(d["ms_search_pass"] * 1e3, 1), round(d["ms_nosearch_pass"] * 1e3, 1))
PY
}
for rep in 1 2 3; do
  for name in none node share sharesmt ccd; do
    pre=""; [ $name != none ] && pre="taskset -c $(grep "^$name " $O/sets.txt | cut -d' ' -f2)"
    $pre timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scans 0 --no-extra-legs > $O/bench20_${name}_$rep.json 2> $O/bench20_${name}_$rep.err
    echo "driver cmd, $name, rep $rep"; line $O/bench20_${name}_$rep.json
    $pre timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs > $O/bench300_${name}_$rep.json 2> $O/bench300_${name}_$rep.err
    echo "300 steps, $name, rep $rep"; line $O/bench300_${name}_$rep.json
  done
done
exit 0
