#!/usr/bin/env python
"""Developer tool: the figures DESIGN.md / README.md quote from a round's evidence files (profiles/rNN_*), recomputed from the files
themselves -- k_pass / k_fit times and roofline fractions per config, counter traffic and its ratio to the algorithmic bytes, L2 hit
share, issue counters, and the bench lines' headline fields.

    python tools/evidence_numbers.py [05]
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "05"
P = os.path.join(ROOT, "profiles")
N = {2: 100_000, 3: 60_000, 4: 130_000, 5: 200_000}


def kstats(cfg):
    out = {}
    with open(os.path.join(P, f"r{RND}_kernel_stats_config{cfg}.csv")) as f:
        for r in csv.DictReader(f):
            m = re.search(r"(k_[a-z_0-9]+(?:<[0-9a-z, ]+>)?)", r["Name"])
            if m:
                out[m.group(1)] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3)
    return out


def pmc(path):
    out = {}
    if not os.path.exists(path):
        return out
    with open(path) as f:
        for r in csv.DictReader(f):
            out[(r["kernel"], r["counter"])] = (int(r["dispatches"]), float(r["mean"]), float(r["sum"]))
    return out


def line(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


for cfg in (2, 3, 4, 5):
    ks = kstats(cfg)
    kp = next(v for k, v in ks.items() if k.startswith("k_pass"))
    kf = next((v for k, v in ks.items() if k.startswith("k_fit<1, false, 2>")), None)
    alg = 117 * N[cfg]
    gbs = alg / (kp[1] * 1e-6) / 1e9
    b = line(os.path.join(P, f"r{RND}_bench_config{cfg}.json"))
    ro = b.get("roofline", {})
    print(f"config {cfg}: k_pass {kp[1]:.2f} us ({kp[0]}; {kp[2]:.1f} / {kp[3]:.0f}) [bench {ro.get('avg_kernel_us')}]  alg {alg / 1e6:.1f} MB  {gbs:.0f} GB/s  "
          f"{100 * gbs / 8000:.2f} %   k_fit {kf[1] if kf else float('nan'):.2f} us")
    s = pmc(os.path.join(P, f"r{RND}_pmc_summary_config{cfg}.csv"))
    kpn = next((k for (k, c) in s if k.startswith("k_pass")), None)
    if kpn:
        fe, wr = s[(kpn, "FETCH_SIZE")][1], s[(kpn, "WRITE_SIZE")][1]
        tot = (2 * fe + wr) * 1024
        kfn = next((k for (k, c) in s if k.startswith("k_fit<1, false, 2>")), None)
        print(f"   traffic 2xFETCH+WRITE = {tot / 1e6:.1f} MB (FETCH {fe / 1024:.2f} MiB, WRITE {wr / 1024:.2f} MiB)  ratio {tot / alg:.2f}x"
              + (f"   k_fit: 2x{s[(kfn, 'FETCH_SIZE')][1] / 1024:.2f} + {s[(kfn, 'WRITE_SIZE')][1] / 1024:.2f} MiB" if kfn else ""))
    l2 = pmc(os.path.join(P, f"r{RND}_pmc_l2_config{cfg}.csv"))
    kpn = next((k for (k, c) in l2 if k.startswith("k_pass")), None)
    if kpn:
        h, m = l2[(kpn, "TCC_HIT_sum")][1], l2[(kpn, "TCC_MISS_sum")][1]
        print(f"   L2 hit share {100 * h / (h + m):.1f} %")
    print(f"   bench: value {b['value']:.0f}  resident {b.get('device_resident_scans_per_s')}  ms/step {b['ms_per_step']}  search/nosearch pass "
          f"{b['ms_search_pass'] * 1e3:.1f} / {b['ms_nosearch_pass'] * 1e3:.1f} us  first/later {ro.get('first_search_us')} / {ro.get('later_search_us')}  "
          f"frac {ro.get('frac')}  traffic {ro.get('traffic')}  repeats {(b.get('value_repeats') or {}).get('all')}")
    if cfg == 3:
        print("   map change kernels:", {k: round(v[1], 2) for k, v in ks.items() if k in ("k_add_resolve", "k_ins_sort_small", "k_brick_rewrite", "k_brick_rewrite_heads", "k_nn_gather", "k_cls_compact",
                                                                                           "k_far_search", "k_far_nearest", "k_mi_classify", "k_add_insert", "k_map_publish")},
              " mi call ms", b.get("ms_map_incremental_call_per_scan"))
    if cfg == 2:
        cb = b.get("cpu_baseline") or {}
        print(f"   cpu baseline {cb.get('value')} scans/s ({cb.get('cores')} threads) => {cb.get('speedup_vs_cpu')}x; sweep best {(cb.get('best_of_thread_sweep') or {}).get('value')} "
              f"({(cb.get('best_of_thread_sweep') or {}).get('cores')} threads) => {(cb.get('best_of_thread_sweep') or {}).get('speedup_vs_cpu')}x")
        mi = b.get("map_incremental") or {}
        print(f"   map_incremental side leg: {mi.get('ms_per_scan')} ms per scan, changes {mi.get('changes')}")
        sq = pmc(os.path.join(P, f"r{RND}_pmc_sq_config2.csv"))
        kpn = next((k for (k, c) in sq if k.startswith("k_pass")), None)
        if kpn:
            iv, av, wv = sq[(kpn, "SQ_INSTS_VALU")][1], sq[(kpn, "SQ_ACTIVE_INST_VALU")][1], sq[(kpn, "SQ_WAVES")][1]
            print(f"   SQ: INSTS_VALU {iv / 1e6:.3f} M per launch / {wv:.0f} waves = {iv / wv:.0f} per wave; ACTIVE_INST_VALU {av / 1e6:.2f} M = {av / 1024 / 1e3:.1f} k per SIMD "
                  f"~ {av / 1024 * 4 / 2.4e3:.1f} us of issue; SALU {sq[(kpn, 'SQ_INSTS_SALU')][1] / wv:.0f} per wave")
        tcp = pmc(os.path.join(P, f"r{RND}_pmc_tcp_config2.csv"))
        kpn = next((k for (k, c) in tcp if k.startswith("k_pass")), None)
        if kpn:
            print(f"   TCP accesses {tcp[(kpn, 'TCP_TOTAL_CACHE_ACCESSES_sum')][1] / 1e6:.2f} M per launch = {tcp[(kpn, 'TCP_TOTAL_CACHE_ACCESSES_sum')][1] / N[2]:.1f} per query")
d = line(os.path.join(P, f"r{RND}_bench_driver_cmd_config2.json"))
ro = d.get("roofline", {})
print(f"driver command: value {d['value']:.0f}  resident {d.get('device_resident_scans_per_s')}  search/nosearch {d['ms_search_pass'] * 1e3:.1f} / {d['ms_nosearch_pass'] * 1e3:.1f}  "
      f"k_pass {ro.get('avg_kernel_us')} first/later {ro.get('first_search_us')} / {ro.get('later_search_us')} frac {ro.get('frac')}  repeats {(d.get('value_repeats') or {}).get('all')}  "
      f"cpu {(d.get('cpu_baseline') or {}).get('value')} => {(d.get('cpu_baseline') or {}).get('speedup_vs_cpu')}x  mi {(d.get('map_incremental') or {}).get('ms_per_scan')} "
      f"{(d.get('map_incremental') or {}).get('changes')}  prelaunched {d.get('prelaunched_nosearch_passes')}")
e = line(os.path.join(P, f"r{RND}_bench_config2_exchanges_one_rank.json"))
for k in ("shard_mode", "other_exchange"):
    x = e.get(k) or {}
    print(k, (x.get("collective") or "")[:30], x.get("value"), x.get("ms_search_pass"), x.get("ms_nosearch_pass"))
print("plain", e.get("value"), e.get("ms_search_pass"), e.get("ms_nosearch_pass"))
