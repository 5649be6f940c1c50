#!/usr/bin/env python
"""bench.py -- scans/s of the FAST-LIO2 measurement update (h_share_model + iterated ESKF) on MI355X.

A "step" is what the node does per scan between receiving feats_down_body and having the posterior
(src/laserMapping.cpp:904-969): the scan is handed over as a HOST buffer (page-locked, flh_host_alloc), staged to
the device (H2D + re-stride + Morton sort, flh_scan_stage_async on the handle's staging thread) WHILE the previous
scan's update runs, then one full update_iterated_dyn_share_modified(): up to 4 h_share_model evaluations (2 of them
with the 5-NN search on this workload) plus the host-side 23x23 algebra.  `value` is the steady-state rate of that
pipeline over >= 100 distinct scans (BASELINE.json configs[1]: 100k-point Avia scan vs 5M-point map); the rate with
the scans already resident in HBM is reported beside it as `device_resident_scans_per_s`.  For --config 3 (scan
stream with incremental map inserts) a step also runs map_incremental (src/laserMapping.cpp:427-474) on the device map.

  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0: the contract fields, `roofline` (HIP events on the handle's stream inside the timed
region; `traffic` from the committed PMC summary of the same command), `cpu_baseline` (the oracle's restated
reference path on a bounded sample, at the reference's 3 OpenMP threads and at all host cores), and -- measured after
the timed region, never part of `value` -- `two_streams_per_gpu`, `map_incremental`, `scan_front_end`.  With N > 1 `value` is the
sharded path (one scan's points split over the ranks -- config 5: the map partitioned -- the ranks' normal equations meeting per
pass as peer-written granules, --exchange peer, or through an RCCL all-reduce, --exchange rccl; the other one timed beside it as
`other_exchange`), the N independent replicas a labelled sub-field.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from fast_lio_amd import capi, synth  # noqa: E402

CONFIGS = {
    # BASELINE.json configs[i]: (M map points, N scan points, sensor)
    1: (200_000, 20_000, "avia"),
    2: (5_000_000, 100_000, "avia"),
    3: (10_000_000, 60_000, "velodyne"),
    4: (20_000_000, 130_000, "ouster64"),
    5: (50_000_000, 200_000, "mid360"),
}
ALG_BYTES_SEARCH = 117  # 16 (query) + 5*16 (neighbours) + 5*4 (index write) + 1 (flag)
ALG_BYTES_NOSEARCH = 97  # 16 + 5*16 (cached neighbours) + 1
PROFILED = bool(os.environ.get("ROCP_TOOL_LIBRARIES"))  # running under rocprofv3
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy)
RING = 4                 # staging slots cycled by the pipelined loop


_JSON_FD = None  # the process's original stdout once keep_stdout_for_the_line() has run


def keep_stdout_for_the_line():
    """The contract is ONE JSON line on stdout.  Libraries write there too -- RCCL prints its version banner through C stdio, which
    is flushed at exit, i.e. BEHIND the line -- so the measuring process points fd 1 at stderr and keeps the original for emit()."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(d):
    data = (json.dumps(d) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, data)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def pmc_traffic(args):
    """HBM bytes per search pass from the committed rocprofv3 --pmc summary of this round (profiles/, made by
    tools/pmc_summary.py from separate FETCH_SIZE and WRITE_SIZE passes of this same command).  FETCH_SIZE/WRITE_SIZE
    are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled (MI355X_MICROARCH.md, HBM
    section).  Only valid for the workload AND the code the summary was taken on: the summary's side file records the hash of the
    library's sources (tools/src_hash.py) and the first-stage setting; a mismatch with the running code drops the figure."""
    import csv
    import glob

    found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_pmc_summary_config{args.config}.csv")))
    if not found or args.lpq != 4 or args.cell != 1.5 or args.index_cache == 0:  # (the summary was taken on the default neighbour cache)
        return None
    path = found[-1]  # the latest round's
    meta = {}
    try:
        meta = json.load(open(path[:-4] + ".meta.json"))
    except (OSError, ValueError):
        pass
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from src_hash import src_hash

        here = src_hash()
    except Exception:  # noqa: BLE001
        here = None
    # the same code: the sources the summary was taken on, or later sources whose DEVICE code was shown to be identical kernel for
    # kernel (tools/asm_equiv.py --bless: experiment blocks compiled out of the product change the hash, not the kernels)
    same_code = bool(meta) and here is not None and (meta.get("src_hash") == here or here in meta.get("device_code_identical_src_hashes", []))
    if not same_code or int(meta.get("pass_kernel", -2)) != int(args.pass_kernel):
        return {"stale": True, "source": os.path.relpath(path, ROOT), "summary_src_hash": meta.get("src_hash"), "running_src_hash": here}
    fetch, write, calls = {}, {}, {}
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["kernel"]
            if not (k.startswith("k_search") or k.startswith("k_pass")):
                continue
            if r["counter"] == "FETCH_SIZE":
                fetch[k] = float(r["sum"])
                calls[k] = int(r["dispatches"])
            elif r["counter"] == "WRITE_SIZE":
                write[k] = float(r["sum"])
    import re

    second = [k for k in calls if re.match(r"k_search_ring<\d+, 2,", k) or k.startswith("k_pass")]
    if not second or not fetch:
        return None
    passes = sum(calls[k] for k in second)  # every search pass launches k_pass (three-launch pass: the second stage) exactly once
    total = (2.0 * sum(fetch.values()) + sum(write.values())) * 1024.0 / passes
    return {"bytes_per_search_pass": int(total), "src_hash": meta.get("src_hash"), "commit": meta.get("commit"),
            "source": os.path.relpath(path, ROOT) + " (FETCH_SIZE x2 + WRITE_SIZE, KiB)"}


def _cpulist(text):
    out = []
    for part in text.strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            out += list(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out


def _all_tasks_affinity(cpus):
    """The affinity of every thread this process has (new threads inherit their creator's)."""
    for t in os.listdir("/proc/self/task"):
        try:
            os.sched_setaffinity(int(t), cpus)
        except OSError:
            pass


def binding_cpus(mode, bdf, node, local, gpus, sib, l3_of):
    """Which CPUs `mode` gives the process of GPU `bdf` (pure: tests/test_bench_binding.py).  local: the allowed CPUs of the GPU's
    NUMA node; gpus: the node's GPUs in PCI order; sib[c]: the hardware threads of c's core; l3_of(c): the CPUs behind c's L3."""
    cpus, note = sorted(local), f"NUMA node {node} of GPU {bdf}"
    if mode not in ("share", "l3", "l3smt"):
        return cpus, note
    phys = [c for c in sorted(local) if sib[c][0] == c]
    if bdf not in gpus or len(phys) < 4 * len(gpus):
        return cpus, note
    per = len(phys) // len(gpus)
    i = gpus.index(bdf)
    share = phys[i * per:(i + 1) * per]
    note = f"GPU {bdf}: share {i + 1} of {len(gpus)} of NUMA node {node}'s {len(phys)} physical cores"
    if mode in ("l3", "l3smt"):  # the cores of the share that sit behind ONE L3 (a CCD): the threads' shared lines stay in it
        l3 = set(l3_of(share[0]))
        ccd = [c for c in share if c in l3]
        if len(ccd) >= 4:
            share = ccd
            note += f", the {len(ccd)} of them behind one L3"
    if mode == "l3":  # one hardware thread per core: two of the process's busy threads never share a core
        return sorted(share), note
    allowed = set(local)
    return sorted({t for c in share for t in sib[c] if t in allowed}), note + ", with SMT siblings"


def bind_to_gpu(device_index, mode):
    """One process per GPU, bound to its GPU's share of the host (what `numactl` / the launcher's binding does in a deployment).
    The update's thread spins on granules the GPU writes over PCIe, the staging thread and the HIP runtime's own threads talk to it
    through shared cache lines: left to the scheduler on a two-socket host they end up on different sockets, behind different L3s
    or on the two hardware threads of one core, and the pipelined loop loses 3-9 % (profiles/r06_call36/ ... r06_call38/).
    The share: the physical cores of the GPU's NUMA node divided among that node's GPUs in PCI order (so eight such processes on
    an 8-GPU host do not overlap).  Same box, alternating (profiles/r06_call38/): the driver's command 7 649 scans/s unbound,
    7 893 on the share, 8 042 on the share's cores behind ONE L3 without their SMT siblings; 300 steps 7 858 / 8 435 / 8 592 -- and
    only the last never fell into the slow mode (whole 37-ms regions 13-17 % low) that the others show now and then.
    mode: "l3" (default: the cores of the share behind ONE L3, one hardware thread each) | "l3smt" (with the SMT siblings) |
    "share" (the whole share, with SMT siblings) | "node" | "off".  Returns what was done (for the line's config) or None;
    never raises."""
    if mode == "off" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        import glob

        import torch

        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        dev = f"/sys/bus/pci/devices/{bdf}"
        node = int(open(dev + "/numa_node").read())
        allowed = os.sched_getaffinity(0)
        local = [c for c in _cpulist(open(dev + "/local_cpulist").read()) if c in allowed]
        if not local:
            return None
        gpus, sib, l3_of = [], {}, None
        if mode in ("share", "l3", "l3smt"):
            for d in sorted(glob.glob("/sys/bus/pci/devices/*")):
                try:
                    if (open(d + "/vendor").read().strip() == "0x1002" and open(d + "/class").read().strip()[:6] in ("0x1200", "0x0302", "0x0380")
                            and int(open(d + "/numa_node").read()) == node):
                        gpus.append(os.path.basename(d))
                except (OSError, ValueError):
                    pass
            sib = {c: _cpulist(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read()) for c in local}
            l3_of = lambda c: _cpulist(open(f"/sys/devices/system/cpu/cpu{c}/cache/index3/shared_cpu_list").read())  # noqa: E731
        cpus, note = binding_cpus(mode, bdf, node, local, gpus, sib, l3_of)
        before = sorted(allowed)
        _all_tasks_affinity(cpus)
        return {"mode": mode, "cpus": len(cpus), "first_cpu": cpus[0], "note": note, "_restore": before}
    except Exception as e:  # noqa: BLE001 -- no /sys, a container without the files, an old torch: run unbound
        log(f"[bench] host binding skipped: {e!r}")
        return None


def load_scene(args, M):
    """The seeded scene; the big ones (>= 10M points: tens of seconds of numpy) are kept in --cache-dir between runs."""
    seed = synth.CONFIG_SEED_BASE + args.config
    path = os.path.join(args.cache_dir, f"scene_cfg{args.config}_M{M}.npz") if (args.cache_dir and M >= 10_000_000) else None
    if path and os.path.exists(path):
        try:
            z = np.load(path)
            sc = synth.Scene(L=float(z["L"]), walls=z["walls"], seed=seed)
            sc.map_xyz = np.ascontiguousarray(z["map_xyz"])
            if sc.map_xyz.shape == (M, 3):
                return sc
        except Exception as e:  # noqa: BLE001
            log(f"[bench] cache {path} unreadable ({e!r}): regenerating")
    sc = synth.make_scene(M, seed)
    if path:
        try:
            os.makedirs(args.cache_dir, exist_ok=True)
            tmp = path + f".tmp{os.getpid()}.npz"
            np.savez(tmp, L=sc.L, walls=sc.walls, map_xyz=sc.map_xyz)
            os.replace(tmp, path)
        except OSError as e:
            log(f"[bench] could not write {path}: {e!r}")
    return sc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--scans", type=int, default=128, help="distinct seeded scans cycled through (>= 100 by default)")
    ap.add_argument("--mode", default="auto", choices=["auto", "streams", "shard", "partition"],
                    help="multi-GPU: shard = ONE scan's points split over the ranks, the ranks' normal equations summed per "
                         "pass (--exchange; strong scaling); partition = the map cut into one slab per rank, whole scan on every "
                         "rank (BASELINE configs[4]); streams = one independent scan stream per rank, replicated map, nothing "
                         "exchanged (replicas).  auto = one GPU: streams; N > 1: shard (config 5: partition) as the headline "
                         "value, the replicas as a sub-field")
    ap.add_argument("--lpq", type=int, default=4)
    ap.add_argument("--cell", type=float, default=1.5)
    ap.add_argument("--pass-kernel", type=int, default=-1,
                    help="flh_config.pass_kernel: -1 / 1 = a searching pass is ONE launch (the library's default), 0 = the three-launch pass")
    ap.add_argument("--exchange", default="peer", choices=["peer", "rccl"],
                    help="N > 1: how the ranks' normal equations meet -- peer = granules written into every rank's pinned buffer "
                         "(no collective, no extra launch; the default), rccl = ncclAllReduce of the 16x16 block on the device + a "
                         "publish kernel.  The other one is timed as a side leg")
    ap.add_argument("--sort", type=int, default=1, help="Morton-order the scan at staging (0 = keep input order)")
    ap.add_argument("--extrinsic-est", type=int, default=0)
    ap.add_argument("--plane-cache", type=int, default=-1, help="flh_config.plane_cache (-1 = the library's default: on)")
    ap.add_argument("--ring", type=int, default=RING,
                    help="staging slots cycled by the pipelined loop; with 3 or more, two scans are staged ahead (on two lanes), with 2 one")
    ap.add_argument("--prelaunch", type=int, default=-1, help="flh_config.prelaunch (-1 = the library's default: on; 0 = every pass is launched when its state is known)")
    ap.add_argument("--index-cache", type=int, default=-1, help="flh_config.index_cache (-1 = default: the neighbour cache holds map indices; 0 = coordinates)")
    ap.add_argument("--stage-sort", type=int, default=-1, help="flh_config.stage_sort (-1 / 1 = the library's own two staging kernels; 0 = restride + vendor radix sort + gather)")
    ap.add_argument("--plane-fit-dtype", type=int, default=0,
                    help="1 = the fp16 plane-fit ABLATION of BASELINE configs[4] (not bit-exact, never a parity claim)")
    ap.add_argument("--timing-samples", type=int, default=16,
                    help="searching evaluations of the timed region whose kernels carry HIP events (start / stop time stamps of "
                         "the kernels themselves); at 20 steps (40 searches) every fifth one: 8 samples, both kinds of search")
    ap.add_argument("--event-stride", type=int, default=0,
                    help="developer: sample every n-th searching evaluation with events, ALSO under a profiler (0 = the default "
                         "sampling, none under a profiler)")
    ap.add_argument("--no-extra-legs", action="store_true", help="only the headline + roofline (profiling runs)")
    ap.add_argument("--repeats", type=int, default=4,
                    help="after the contract's timed region, repeat it this many times on the same stream -> value_repeats (median/min/max)")
    ap.add_argument("--cpu-scans", type=int, default=96,
                    help="upper bound of the scans timed on the CPU oracle at --cpu-threads (it stops after ~12 s; 0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=3, help="OpenMP threads (reference MP_PROC_NUM = 3)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo for debugging")
    ap.add_argument("--single-device", type=int, default=0, help="debug: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--leg", default="", help="internal: run only the named group of side legs (used by the child process)")
    ap.add_argument("--two-streams", action="store_true", help="side legs: also time two scan streams on one GPU")
    ap.add_argument("--cache-dir", default=os.path.join(ROOT, ".bench_cache"),
                    help="where generated scans + priors are kept between runs ('' = do not cache)")
    ap.add_argument("--bind", default="l3", choices=["l3", "l3smt", "share", "node", "off"],
                    help="host binding of this process (one process per GPU): l3 = the cores behind one L3 inside the GPU's share of "
                         "its NUMA node's cores, one hardware thread each (default); l3smt = with their SMT siblings; share = that "
                         "whole share; node = the whole node; off = leave it to the scheduler")
    ap.add_argument("--diag-fresh-repeats", action="store_true",
                    help="diagnostic (tools/r06_call28.sh): the repeats of the contract's region run scans nobody has seen yet "
                         "instead of the region's own (is the region slower because its scans are new, or because it is first?)")
    ap.add_argument("--diag-staging", action="store_true",
                    help="diagnostic: per region, the library's staging / activation counters (flh_debug_stage_stats) -> staging_diag")
    ap.add_argument("--diag-pretouch", type=int, default=0,
                    help="diagnostic: before the warm-up, 1 = every scan's host buffer crosses PCIe once (to a scratch slot), "
                         "2 = every scan is searched once against the map (resident): which first touch costs the contract's region?")
    ap.add_argument("--fresh-scans", action="store_true",
                    help="generate the scans in this process even when --cache-dir holds them (the condition of the two device faults "
                         "of rounds 2 and 3: 16 host threads allocating right before the GPU work; tools/fault_hunt.sh)")
    ap.add_argument("--in-process", action="store_true",
                    help="one GPU: do the GPU work in this process (default: in a child that is started once more if it dies, "
                         "so that a transient device fault costs a retry and not the line); profilers want this flag")
    ap.add_argument("--force-shard-leg", action="store_true",
                    help="debug: run the shard / partition leg on ONE rank too (a one-rank RCCL communicator), to exercise its code")
    args = ap.parse_args()
    if args.leg == "extras":
        return extra_legs(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and not args.in_process and args.leg != "main":
        return run_main_in_child()
    keep_stdout_for_the_line()
    local_rank = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log(f"[bench] WORLD_SIZE={world} != --gpus {args.gpus}; using WORLD_SIZE")
    G = world

    import torch

    if not torch.cuda.is_available() or not capi.device_available():
        log("bench.py needs a GPU (the product has no CPU fallback)")
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dist = None
    if G > 1:
        import torch.distributed as dist

        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    mode = args.mode
    if mode == "auto":
        # one GPU: the pipelined scan stream.  N > 1: BASELINE's split -- ONE scan's points sharded over the ranks (configs with a
        # map <= 20M points: map replicated) or the map partitioned (config 5), the 16x16 normal-equation block all-reduced by RCCL
        # inside every pass.  N independent replicas (no collective) are reported beside it, never as `value`.
        mode = "streams" if G == 1 else ("partition" if args.config == 5 else "shard")
    run_shard_leg = (G > 1 or args.force_shard_leg) and args.mode in ("auto", "shard", "partition")
    run_replica_leg = G > 1 and args.mode == "auto"

    from fast_lio_amd import dist as fdist

    M, N, sensor = CONFIGS[args.config]
    ext = bool(args.extrinsic_est)
    with_map_inserts = args.config == 3  # "Velodyne scan stream ... incremental map inserts"
    t0 = time.time()
    scene = load_scene(args, M)
    S = max(1, args.scans)  # >= 100 distinct seeded scans whatever --steps is; they are cycled through

    class _Scan:  # what the legs need of a synth.Problem
        def __init__(self, body, x_prior):
            self.body, self.x_prior = body, x_prior

    def gen(seed_base, count):
        """`count` seeded scans + propagated priors.  Generating a scan ray-casts against every wall of the scene (tens of
        seconds per scan for the 20M / 50M-point scenes), so the result is kept in --cache-dir (deterministic: same seeds, same
        bytes) -- the rocprofv3 passes of the same command do not generate it again."""
        from concurrent.futures import ThreadPoolExecutor

        cache = None
        if args.cache_dir and not args.fresh_scans:
            os.makedirs(args.cache_dir, exist_ok=True)
            cache = os.path.join(args.cache_dir, f"scans_cfg{args.config}_n{count}_b{seed_base}.npz")
            if os.path.exists(cache):
                try:
                    z = np.load(cache)
                    if z["body"].shape == (count, N, 3):
                        pr = [_Scan(np.ascontiguousarray(z["body"][i]), z["x_prior"][i]) for i in range(count)]
                        return pr, [(np.ascontiguousarray(z["x"][i]), np.ascontiguousarray(z["P"][i])) for i in range(count)]
                except Exception as e:  # a torn file: generate again
                    log(f"[bench] cache {cache} unreadable ({e!r}): regenerating")
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:  # numpy releases the GIL in the heavy parts
            full = list(ex.map(lambda s_: synth.make_problem(M, N, sensor, cfg=args.config, scan_seed=seed_base + s_, scene=scene),
                               range(count)))
        pr = [_Scan(p.body, p.x_prior) for p in full]
        pri = [synth.propagate_prior_cov(capi.predict_fn, p.x_prior) for p in pr]
        pri = [(np.ascontiguousarray(x, np.float64), np.ascontiguousarray(P, np.float64)) for x, P in pri]
        if cache:
            try:
                tmp = cache + f".tmp{os.getpid()}.npz"
                np.savez(tmp, body=np.stack([p.body for p in pr]), x_prior=np.stack([p.x_prior for p in pr]),
                         x=np.stack([x for x, _ in pri]), P=np.stack([P for _, P in pri]))
                os.replace(tmp, cache)
            except OSError as e:
                log(f"[bench] could not write {cache}: {e!r}")
        return pr, pri

    # streams: every rank follows its own scan stream; the shard leg uses scans common to all ranks
    probs, priors = gen(1000 * rank if G > 1 else 0, S)
    S_sh = min(S, 8)
    sh_probs, sh_priors = (gen(0, S_sh) if (rank != 0 and G > 1) else (probs[:S_sh], priors[:S_sh])) if run_shard_leg else (None, None)
    # (after the scans' generation -- sixteen worker threads -- and before the library's threads and the page-locked buffers exist)
    binding = bind_to_gpu(local_rank, args.bind)
    # the scans as the node would hold them: host buffers (page-locked so that the DMA engine reads them where they lie)
    bodies = []
    for p in probs:
        a = capi.pinned_empty((N, 3), np.float32)
        a[:] = p.body
        bodies.append(a)
    if rank == 0:
        log(f"[bench] config {args.config}: M={M} N={N} sensor={sensor} scans={S} ranks={G} mode={args.mode} "
            f"gen {time.time() - t0:.1f}s")

    h = capi.Handle(cell_size=args.cell, lanes_per_query=args.lpq, device=local_rank, sort_queries=args.sort,
                    pass_kernel=args.pass_kernel, plane_cache=args.plane_cache, plane_fit_dtype=args.plane_fit_dtype,
                    prelaunch=args.prelaunch, index_cache=args.index_cache, stage_sort=args.stage_sort)
    t0 = time.time()
    h.map_build(scene.map_xyz)
    t_build = time.time() - t0
    lo, hi = fdist.shard_bounds(N, rank, G)
    kf = capi.Esekf(h, max_iter=3, extrinsic_est_en=ext)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    class Acc:
        def __init__(self, rs=None):
            self.passes = self.searches = 0
            self.ms_s = self.ms_n = self.ms_mi = 0.0
            self.n_s = self.n_n = 0
            if rs is not None:
                self.passes, self.searches = int(rs.passes), int(rs.searches)
                self.ms_s, self.n_s = float(rs.ms_search_passes), int(rs.n_search_passes)
                self.ms_n, self.n_n = float(rs.ms_nosearch_passes), int(rs.n_nosearch_passes)
                self.ms_mi = float(rs.ms_map_incremental)

    staging_diag = []  # (--diag-staging)

    def run(kfx, hx, jobs, n_warm, n_steps, first=0):
        """W untimed warm-up scans, then EXACTLY n_steps scans inside one native call (flh_esekf_run_scans: the node's main
        loop, scan i+1 staged while scan i updates) bracketed by barrier + device synchronisation; max over ranks."""
        # Everything that costs the HOST time and leaves the device idle -- Python's cyclic GC (kept out of the timed region, as
        # timeit does; after scene generation a collection takes tens of milliseconds), the creation of the event pool -- happens
        # BEFORE the warm-up, so that the W warm-up scans are immediately followed by the K timed ones: round 5's call 2 showed the
        # contract's first region 7 % below its own repeats (profiles/r05_call2/: 6 515 vs 6 970-7 030 scans/s) with the idle gap
        # between warm-up and measurement, which hands the first timed scans a device that has clocked down again.
        gc.collect()
        gc.disable()
        # The kernels of a sampled evaluation carry HIP events (read after the timed region); that costs the host ~10 us per
        # sampled evaluation, so only SEARCHING evaluations are sampled -- the roofline is the search's -- every n-th of them,
        # n odd: a scan's first and later searches alternate, an odd stride samples both kinds alike
        stride = max(5, (n_steps * 2) // max(args.timing_samples, 8))  # at most one searching evaluation in five carries events
        # under a profiler (rocprofv3 sets ROCP_TOOL_LIBRARIES) no events: its trace IS the kernel timing and stays free of the
        # events' cost (--event-stride forces them: DESIGN.md 6, the fault hunt)
        hx.set_timing_sampling(args.event_stride if args.event_stride > 0 else (0 if PROFILED else stride + 1 - (stride & 1)), True)
        # the stream does not stop at the boundary of the timed region: the first timed scan is staged while the last
        # warm-up scan updates, exactly as every later scan is staged while its predecessor updates
        kfx.run_scans(jobs, first, n_warm, ring=args.ring, map_incremental=with_map_inserts, stage_next=True)
        sync()
        hx.counters(reset=True)  # (the warm-up's samples are dropped)
        if args.diag_staging:
            hx.stage_stats(reset=True)
        t1 = time.perf_counter()
        rs = kfx.run_scans(jobs, first + n_warm, n_steps, ring=args.ring, map_incremental=with_map_inserts, first_staged=n_warm > 0)
        sync()
        dt_ = time.perf_counter() - t1
        if args.diag_staging:
            sd = hx.stage_stats()
            nj, na = max(sd["jobs"], 1.0), max(sd["activations"], 1.0)
            staging_diag.append({"scans_per_s": round(n_steps / dt_, 1), "ms_search_pass": round(float(rs.ms_search_passes) / max(int(rs.n_search_passes), 1), 4),
                                 "ms_nosearch_pass": round(float(rs.ms_nosearch_passes) / max(int(rs.n_nosearch_passes), 1), 4),
                                 "stage_enq_us": round(sd["enq_us"] / nj, 2), "stage_enq_max_us": round(sd["enq_max_us"], 1),
                                 "h2d_wait_us": round(sd["h2d_wait_us"] / nj, 2), "h2d_wait_max_us": round(sd["h2d_wait_max_us"], 1),
                                 "act_wait_us": round(sd["act_wait_us"] / na, 2), "act_wait_max_us": round(sd["act_wait_max_us"], 1),
                                 "act_event_not_ready": int(sd["act_event_not_ready"]), "act_slot_pending": int(sd["act_slot_pending"]),
                                 "jobs": int(sd["jobs"]), "activations": int(sd["activations"])})
        gc.enable()
        hx.set_timing_stride(0)
        if dist is not None:
            tt = torch.tensor([dt_], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ = float(tt.item())
        c_ = hx.counters()
        c_.update(hx.search_counters())
        return dt_, Acc(rs), c_

    # ---------------- headline leg: the pipelined loop.  Scan i+1 is handed to the staging thread (host buffer -> H2D ->
    # re-stride + Morton sort on the copy stream) before the update of scan i starts; the update waits for ITS staging
    # only.  Every scan of the timed region crosses PCIe inside the timed region.
    jobs_pipe = capi.Esekf.make_jobs(bodies, priors)

    shard_out = None
    if mode in ("shard", "partition") and G == 1 and not args.force_shard_leg:
        mode = "streams"  # one rank: nothing to shard
    replicas_out = None
    repeats = None
    if mode == "streams":
        if args.diag_pretouch:  # (diagnostic only: never part of a reported line's contract)
            for i, b in enumerate(bodies):
                h.scan_upload(b)
                if args.diag_pretouch >= 2:
                    h.eval(np.asarray(priors[i][0], np.float64), True, False)
            sync()
        dt, acc, ctr = run(kf, h, jobs_pipe, args.warmup, args.steps)
        units = args.steps * G
        n_pts = N
        # the contract's fields are THIS region's.  One region of the driver's command is 3 ms on boxes that differ by 20 %: the
        # same region is repeated on the same stream (the scans keep cycling, no new warm-up beyond two scans) so that the line
        # carries a spread beside its one number
        if args.repeats > 0 and not PROFILED:
            vals = [units / dt]
            for r_ in range(args.repeats):
                first_r = (args.warmup + args.steps + r_ * (args.steps + 2)) if args.diag_fresh_repeats else 0
                if first_r + 2 + args.steps > len(bodies):
                    first_r = 0
                dt_r, _a, _c = run(kf, h, jobs_pipe, 2, args.steps, first_r)
                vals.append(units / dt_r)
            sv = sorted(vals)
            repeats = {"regions": len(vals), "median": round(sv[len(sv) // 2], 3), "min": round(sv[0], 3), "max": round(sv[-1], 3),
                       "all": [round(v, 1) for v in vals],
                       "note": "the first entry of `all` is the contract's region (= value); the others repeat it on the same stream"}
    elif run_replica_leg:
        k1 = max(10, min(60, args.steps // 2))
        dt1, _a1, _c1 = run(kf, h, jobs_pipe, max(3, args.warmup // 2), k1)
        replicas_out = {"value": round(k1 * G / dt1, 3), "unit": "scans/s", "steps": k1, "scaling": "weak",
                        "note": f"{G} independent scan streams (one per rank), replicated map, NO collective in the data path: "
                                "replicas, not the sharded path"}
    # ---------------- sharded leg (north_star C1): ONE scan's points split over the ranks Morton-first, map replicated (or,
    # --mode partition / config 5, the map cut into slabs with a halo and every rank holding the whole scan); per pass each
    # rank reduces its part to the 16x16 Gram block in device memory and RCCL sums the blocks INSIDE flh_eval (native
    # call site, no Python and no D2H in the pass); every rank then runs the identical 23x23 solve.
    class ExchangeUnavailable(RuntimeError):
        """the named exchange did not come up on every rank (raised on ALL ranks, after they have agreed on it)"""

    def all_ranks_ok(ok):
        """did this step succeed on EVERY rank?  One all-reduce that every rank reaches: the steps it guards catch their own errors."""
        if dist is None:
            return ok
        t_ = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(t_, op=dist.ReduceOp.MIN)
        return bool(int(t_.item()))

    def why(err):
        """the first rank's error text, known to every rank (called by all ranks after they agreed that something failed)"""
        if dist is None:
            return err or "failed"
        errs = [None] * G
        dist.all_gather_object(errs, err)
        return next((f"rank {r_}: {e_}" for r_, e_ in enumerate(errs) if e_), "failed")

    def shard_leg(exchange, n_warm, n_steps):
        """ONE scan over the ranks with the named exchange; returns (dict for the line, (dt, acc, ctr), points on this rank).
        Setting the exchange up is fallible (a shared segment that cannot be page-locked, an RCCL that is not there) and must not
        leave the ranks in different collectives: every fallible step catches its own error, the ranks then agree, and a failure
        anywhere raises ExchangeUnavailable everywhere."""
        partition = args.mode == "partition" or (args.mode == "auto" and args.config == 5)
        hs, err = None, None
        tok = [None]
        try:
            hs = capi.Handle(cell_size=args.cell, lanes_per_query=args.lpq, device=local_rank, sort_queries=args.sort,
                             pass_kernel=args.pass_kernel, index_cache=args.index_cache, stage_sort=args.stage_sort)
            if rank == 0:  # the token every rank needs: RCCL's unique id / the name of the shared segment
                tok = [capi.rccl_unique_id() if exchange == "rccl" else f"/flh_bench_{os.getpid()}"]
        except Exception as e:  # noqa: BLE001
            err = repr(e)[:300]
        if dist is not None:
            dist.broadcast_object_list(tok, src=0)
        if err is None and tok[0] is None:
            err = "rank 0 could not make the exchange's token"
        if not all_ranks_ok(err is None):  # before the set-up proper: ncclCommInitRank blocks until every rank has joined
            if hs is not None:
                hs.close()
            raise ExchangeUnavailable(why(err))
        try:
            if exchange == "rccl":
                hs.rccl_init_rank(G, tok[0], rank)
            else:
                hs.peer_open(tok[0], G, rank)
        except Exception as e:  # noqa: BLE001
            err = repr(e)[:300]
        if not all_ranks_ok(err is None):
            hs.close()
            raise ExchangeUnavailable(why(err))
        if partition:
            axis, edges = fdist.partition_bounds(scene.map_xyz, G)
            keep = fdist.partition_slab(scene.map_xyz, axis, edges, rank, fdist.HALO_DEFAULT)
            hs.map_build(scene.map_xyz[keep])
            hs.set_owned_interval(axis, edges[rank], edges[rank + 1])
            for s, p in enumerate(sh_probs):
                hs.scan_stage(s, p.body)
            pts_here = N
        else:
            hs.map_build(scene.map_xyz)
            for s, p in enumerate(sh_probs):
                hs.scan_stage(s, np.ascontiguousarray(p.body[fdist.morton_shard(p.body, rank, G)]))
            pts_here = hi - lo
        kfs = capi.Esekf(hs, max_iter=3, extrinsic_est_en=ext)
        jobs_sh = capi.Esekf.make_jobs([np.zeros((1, 3), np.float32)] * S_sh, sh_priors, slots=list(range(S_sh)))
        dt2, acc2, ctr2 = run(kfs, hs, jobs_sh, n_warm, n_steps)
        # every rank must have produced the same posterior
        xs = [kfs.get_x()] * G
        if dist is not None:
            dist.all_gather_object(xs, kfs.get_x())
        agree = float(max(np.abs(np.asarray(x_) - np.asarray(xs[0])).max() for x_ in xs))
        d = {"value": round(n_steps / dt2, 3), "unit": "scans/s", "steps": n_steps, "ms_per_step": round(dt2 / n_steps * 1e3, 4),
             "ms_per_iekf_pass": round(dt2 / max(acc2.passes, 1) * 1e3, 4),
             "ms_search_pass": round(acc2.ms_s / max(acc2.n_s, 1), 4), "ms_nosearch_pass": round(acc2.ms_n / max(acc2.n_n, 1), 4),
             "layout": ("map partitioned into slabs (+%.2f m halo), whole scan on every rank, queries owned by position" % fdist.HALO_DEFAULT
                        if partition else "scan sharded Morton-first, map replicated"),
             "points_per_rank": pts_here, "map_points_this_rank": hs.M,
             "collective": ("rccl: ncclAllReduce(sum) of the pass's group totals (64 groups x 31 f64; 94 with the extrinsic columns) per pass, issued by flh_eval on the handle's stream, + a publish kernel that adds the groups"
                            if exchange == "rccl" else
                            "peer granules: every rank's group reducers write {value, sequence} granules into every rank's pinned buffer "
                            "(one shared segment); no collective, no extra launch; every host adds (rank, group) in order"),
             "ranks_in_communicator": hs.rccl_size() if exchange == "rccl" else hs.peer_size(),
             "max_abs_state_disagreement_across_ranks": agree, "scaling": "strong"}
        kfs.close()
        hs.close()
        return d, (dt2, acc2, ctr2), pts_here

    other_exchange = None
    if run_shard_leg:
        headline = mode in ("shard", "partition")
        oth = "rccl" if args.exchange == "peer" else "peer"
        k2 = args.steps if headline else max(10, min(60, args.steps // 4))  # (a side leg only with --force-shard-leg)
        w2 = args.warmup if headline else max(3, args.warmup // 4)
        used, res = args.exchange, None
        try:
            res = shard_leg(args.exchange, w2, k2)
        except ExchangeUnavailable as e1:
            # the chosen exchange did not come up on this node: the sharded path is then measured with the other one, and the line says so
            used = oth
            try:
                res = shard_leg(oth, w2, k2)
                res[0]["exchange_fallback"] = f"--exchange {args.exchange} failed ({e1}); measured with {oth}"
            except ExchangeUnavailable as e2:
                if headline:
                    raise RuntimeError(f"sharded leg: neither exchange came up: {e1} / {e2}")
                shard_out = {"error": f"{e1} / {e2}"}
        if res is not None:
            shard_out, (dt2, acc2, ctr2), pts_here = res
            if headline:
                dt, acc, ctr = dt2, acc2, ctr2
                units = args.steps
                n_pts = pts_here
        if used == args.exchange:
            try:  # the other exchange, shorter, for comparison
                other_exchange, _r, _p = shard_leg(oth, max(3, args.warmup // 4), max(10, min(60, args.steps // 2)))
            except ExchangeUnavailable as e:
                other_exchange = {"error": str(e)[:400]}
    value = units / dt
    ms_per_step = dt / args.steps * 1e3

    # ---- roofline of the dominant kernels (the 5-NN search of one pass), HIP events inside the timed region
    roof = None
    one_launch = args.pass_kernel != 0 and args.lpq == 4 and abs(args.cell - 1.5) < 1e-6 and not args.plane_fit_dtype
    if ctr["n_search"] > 0:
        dur_s = ctr["search_ms"] / ctr["n_search"] * 1e-3
        ach = ALG_BYTES_SEARCH * n_pts / dur_s / 1e9
        roof = {"bound": "hbm",
                "kernel": ("k_pass = a whole searching pass in ONE launch: body->world, 5-NN (both stages), plane fit, residual gate, "
                           "Jacobian rows, Gram (f64 MFMA), group sums" if one_launch else
                           "5-NN search of one pass = k_search_ring<4,1> (every query) + k_search_ring<8,2> (the rest, incl. the exact fallback)"),
                "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                "traffic": None, "avg_kernel_us": round(dur_s * 1e6, 2), "alg_bytes_per_launch": ALG_BYTES_SEARCH * n_pts,
                "event_bracket": ("start -> end of k_pass (hipExtLaunchKernelGGL time stamps)" if one_launch else
                                  "start of the first search kernel -> end of the last one (hipExtLaunchKernelGGL time stamps)"),
                "events_sampled": int(ctr["n_search"]),
                "events_sampled_by_kind": {"first_search_of_scan": int(ctr.get("n_first", 0)), "later_search": int(ctr.get("n_later", 0))},
                "first_search_us": round(ctr["first_ms"] / ctr["n_first"] * 1e3, 2) if ctr.get("n_first") else None,
                "later_search_us": round(ctr["later_ms"] / ctr["n_later"] * 1e3, 2) if ctr.get("n_later") else None}
        if ctr["n_fit"] > 0:  # the three-launch pass: the fit kernel behind a sampled search
            fit_s = ctr["fit_ms"] / ctr["n_fit"] * 1e-3
            roof.update({"fit_kernel_us": round(fit_s * 1e6, 2), "fit_events_sampled": int(ctr["n_fit"]),
                         "fit_alg_bytes_per_launch": ALG_BYTES_NOSEARCH * n_pts,
                         "fit_achieved_GBs": round(ALG_BYTES_NOSEARCH * n_pts / fit_s / 1e9, 2),
                         "fit_frac": round(ALG_BYTES_NOSEARCH * n_pts / fit_s / 1e9 / HBM_PEAK_GBS, 5)})

    if with_map_inserts:
        acc_mi = acc.ms_mi / max(args.steps, 1)
    out = {
        "metric": "scans/sec + ms/IEKF-iter, 100k-pt scan vs 5M-pt map, 1/2/4/8 MI355X",
        "value": round(value, 3),
        "unit": "scans/s",
        "n_gpus": G,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong" if mode in ("shard", "partition") else "weak",
        "vs_baseline": None,
        "dtype": "f32 (kNN, plane fit) + f64 (transform, Jacobian, normal equations)",
        "data": "synthetic",
        "config": {"workload": f"BASELINE configs[{args.config - 1}]: {sensor} {N}-pt scan vs {M}-pt box-city map, "
                               f"max_iteration=3, R=0.001, extrinsic_est_en={int(ext)}"
                               + (", map_incremental after every update" if with_map_inserts else ""),
                   "timed_region": ("scan handed over as a page-locked host buffer, staged (H2D + re-stride + Morton sort) on a "
                                    "copy stream while earlier scans update (up to two scans staged ahead, on two lanes), then the full iterated update"
                                    if mode not in ("shard", "partition") else "scan shards resident in HBM, full iterated update"),
                   "parallelism": ("1 GPU" if G == 1 else
                                   ((f"map partitioned over {G} ranks (slabs + halo), whole scan on every rank, queries owned by position"
                                     if mode == "partition" else f"scan points sharded over {G} ranks") +
                                    f", normal equations summed per pass ({used if run_shard_leg else args.exchange})"
                                    if mode in ("shard", "partition") else
                                    f"{G} independent scan streams (one per rank), replicated map, no collective in the data path")),
                   "host_binding": ({k: v for k, v in binding.items() if not k.startswith("_")} if binding else "none"),
                   "distinct_scans": S, "staging_ring": args.ring, "cell_size_m": args.cell, "lanes_per_query": args.lpq,
                   "searching_pass": "one launch (k_pass)" if one_launch else "three launches (two search stages + fit)",
                   "plane_cache": args.plane_cache, "plane_fit": "fp16 ABLATION (not bit-exact)" if args.plane_fit_dtype else "fp32 (reference-exact)",
                   "event_reading": "deferred (recorded inside the timed region, read after it)"},
        "ms_per_iekf_pass": round((acc.ms_s + acc.ms_n) / max(acc.passes, 1), 4),
        "ms_search_pass": round(acc.ms_s / max(acc.n_s, 1), 4),
        "ms_nosearch_pass": round(acc.ms_n / max(acc.n_n, 1), 4),
        "passes_per_scan": round(acc.passes / args.steps, 3),
        "searches_per_scan": round(acc.searches / args.steps, 3),
        "map_build_s": round(t_build, 3),
    }
    if repeats is not None:
        out["value_repeats"] = repeats
    if staging_diag:
        out["staging_diag"] = staging_diag  # diagnostic: one entry per region run()
    if with_map_inserts:
        # host time inside flh_map_incremental per scan: enqueueing the classification and the Add_Points work (nobody asks for the
        # two list lengths, so the host does not wait for them when the previous change was of a scan's usual size); the device
        # finishes the change while the host stages / activates the next scan -- all inside ms_per_step
        out["ms_map_incremental_call_per_scan"] = round(acc_mi, 4)
    if G > 1 and dist is not None:
        out["ranks_seen_by_collective"] = int(dist.get_world_size())
        out["ranks_exchanging_normal_equations"] = shard_out.get("ranks_in_communicator") if shard_out else None

    # ---- the same update with the scans already resident in HBM (staged before the timed loop): what round 1 reported
    # as `value`.  Not PCIe-inclusive, hence a sub-field.
    extra = rank == 0 and G == 1 and mode == "streams" and not args.no_extra_legs
    if rank == 0 and G == 1 and mode == "streams":
        Sd = min(S, 32)
        for s in range(Sd):
            h.scan_stage(16 + s, bodies[s])
        jobs_res = capi.Esekf.make_jobs(bodies[:Sd], priors[:Sd], slots=[16 + s for s in range(Sd)])
        dtr, accr, _ = run(kf, h, jobs_res, max(4, args.warmup // 2), args.steps)
        out["device_resident_scans_per_s"] = round(args.steps / dtr, 3)
        out["device_resident_ms_per_step"] = round(dtr / args.steps * 1e3, 4)

    # companion figure: mean map points examined per query by one search pass
    h.enable_stats(True)
    h.set_timing_stride(0 if PROFILED else 1)
    h.scan_upload(bodies[0])
    h.eval(priors[0][0], True, ext)
    cand_per_query = h.timing()["candidates"] / max(N, 1)
    h.enable_stats(False)
    h.set_timing_stride(0)
    instrumented, words = h.debug_bounds()
    if instrumented:  # a -DFLH_BOUNDS developer build (tools/fault_hunt.sh): violations per translation unit {count, site, index, cap, block}
        out["debug_bounds"] = {"kernels": words[0:5], "pass": words[5:10], "mapinc": words[10:15], "scanprep": words[15:20]}
    ps = h.pass_stats()
    out["prelaunched_nosearch_passes"] = dict(h.prelaunch_stats(), setting=args.prelaunch)  # kernels enqueued ahead / handed their state / released unused / given up
    out["config"]["index_cache"] = args.index_cache
    out["config"]["stage_sort"] = args.stage_sort
    out["second_stage_queries_per_search_pass"] = round(ps["second_stage_queries"] / max(ps["search_passes"], 1), 1)
    if roof is not None:
        tr = pmc_traffic(args)
        if tr is not None and not tr.get("stale"):
            roof["traffic"] = tr["bytes_per_search_pass"]
            roof["traffic_source"] = tr["source"]
            roof["traffic_src_hash"] = tr["src_hash"]
            roof["traffic_commit"] = tr["commit"]
        elif tr is not None:
            roof["traffic_note"] = (f"dropped: {tr['source']} was taken on other code (sources {tr['summary_src_hash']}, running "
                                    f"{tr['running_src_hash']})")
        roof["candidates_per_query"] = round(cand_per_query, 2)
        roof["candidate_traffic_GBs"] = round(cand_per_query * 16 * n_pts / (roof["avg_kernel_us"] * 1e-6) / 1e9, 2)
        out["roofline"] = roof
    if shard_out is not None and mode not in ("shard", "partition"):
        out["shard_mode"] = shard_out
    if shard_out is not None and mode in ("shard", "partition"):
        out["sharded_path"] = {k: shard_out[k] for k in ("layout", "points_per_rank", "map_points_this_rank", "collective",
                                                          "ranks_in_communicator", "max_abs_state_disagreement_across_ranks",
                                                          "ms_search_pass", "ms_nosearch_pass")}
        if "exchange_fallback" in shard_out:
            out["sharded_path"]["exchange_fallback"] = shard_out["exchange_fallback"]
    if other_exchange is not None:
        out["other_exchange"] = other_exchange
    if replicas_out is not None:
        out["replicas_no_collective"] = replicas_out

    # ---- CPU baseline: the oracle's restated reference path on this box's host cores (rank 0, N=1 only): a bounded sample
    # (~10-30 s of CPU work) at the reference's own thread count (MP_PROC_NUM = 3, CMakeLists.txt:21-24), and the best of a
    # small thread sweep -- the restated k-d tree path stops scaling long before "all cores" on a many-core host
    if rank == 0 and G == 1 and args.cpu_scans > 0:
        from oracle import pyoracle as po

        if binding:  # the CPU baseline gets the whole host, as before: its OpenMP threads are created from here
            _all_tasks_affinity(binding["_restore"])

        m = po.Map(scene.map_xyz)  # k-d tree build is outside the reference's t_update window too
        ncores = os.cpu_count() or 1

        def cpu_rate(threads, budget_s, max_scans):
            tot, n = 0.0, 0
            while n < max_scans and (n < 2 or tot < budget_s):
                s_ = n % S
                sc = po.Scan(probs[s_].body, nthreads=threads)
                t1 = time.perf_counter()
                sc.update_iterated(m, priors[s_][0], priors[s_][1], extrinsic_est_en=ext)
                tot += time.perf_counter() - t1
                n += 1
            return n / tot, n

        r3, n3 = cpu_rate(args.cpu_threads, 12.0, args.cpu_scans)
        sweep = {}
        for th in (8, 16, 32, 64):
            if th <= ncores:
                sweep[th] = cpu_rate(th, 2.5, max(2, args.cpu_scans // 4))
        best_th = max(sweep, key=lambda k_: sweep[k_][0]) if sweep else None
        out["cpu_baseline"] = {"value": round(r3, 4), "unit": "scans/s", "cores": args.cpu_threads, "kind": "port",
                               "sample": f"{n3} full updates of the same {N}-pt scans vs the same {M}-pt map "
                                         f"(restated reference path: k-d tree 5-NN + plane fit + IEKF, OpenMP "
                                         f"{args.cpu_threads} threads = the reference's MP_PROC_NUM); host has "
                                         f"{ncores} logical cores",
                               "speedup_vs_cpu": round(value / r3, 1)}
        if best_th is not None:
            out["cpu_baseline"]["best_of_thread_sweep"] = {
                "value": round(sweep[best_th][0], 4), "cores": best_th, "speedup_vs_cpu": round(value / sweep[best_th][0], 1),
                "sweep_scans_per_s": {str(k_): round(v_[0], 3) for k_, v_ in sweep.items()},
                "sample": f"{sweep[best_th][1]} updates per thread count"}

    # ---- the legs beside the headline (map_incremental, the raw-scan front end incl. frame_world, optionally two scan
    # streams on one GPU) run in a CHILD process after this one has finished its own GPU work: they are reported beside
    # the contract fields, never part of them, and a failure there must not cost the line
    if extra:
        kf.close()
        h.close()
        kf = h = None
        out.update(run_extra_legs_in_child(args))

    if rank == 0:
        emit(out)
    if kf is not None:
        kf.close()
        h.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_main_in_child(attempts=2):
    """One GPU: the whole measurement runs in a child process (python bench.py <same flags> --leg main); its stderr passes
    through, its JSON line is re-printed here.  A child that dies without a line (a device fault aborts the process) is
    started once more; the second failure is the bench's failure.  Nothing is measured differently: the child is main()."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:]] + ["--leg", "main"]
    big = any(a in ("4", "5") and i > 0 and sys.argv[i] == "--config" for i, a in enumerate(sys.argv[1:]))  # 20M / 50M maps: generation alone takes minutes
    limit = 3000 if big else 800  # a healthy run of the default configuration takes one to three minutes
    rc = 1
    history = []
    for k in range(attempts):
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, timeout=limit)
            rc, out = r.returncode, r.stdout
        except subprocess.TimeoutExpired as e:
            rc, out = -9, (e.stdout or b"")
        history.append(rc)
        line = None
        for ln in out.decode(errors="replace").splitlines():
            if ln.startswith("{"):
                try:
                    d = json.loads(ln)
                    if "metric" in d:
                        line = d
                except ValueError:
                    pass
        if line is not None:  # printed once, after every measurement: it stands even if the process then died while tearing down
            # nothing is hidden: the line says how many attempts it took and how each measuring process ended
            line["measuring_process"] = {"attempts": k + 1, "exit_codes": history}
            if rc != 0:
                log(f"[bench] the measuring process printed its line and then exited with {rc}")
            print(json.dumps(line), flush=True)
            return 0
        log(f"[bench] attempt {k + 1}/{attempts}: the measuring process exited with {rc} and no result line")
        if rc == 3:  # no GPU: a second try changes nothing
            break
    sys.exit(rc if rc > 0 else 1)


def run_extra_legs_in_child(args):
    """python bench.py --leg extras ... in a child process; returns its dict (or an error note)."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--leg", "extras", "--config", str(args.config), "--lpq", str(args.lpq),
           "--cell", str(args.cell), "--pass-kernel", str(args.pass_kernel), "--sort", str(args.sort),
           "--extrinsic-est", str(args.extrinsic_est), "--steps", str(args.steps), "--index-cache", str(args.index_cache), "--stage-sort", str(args.stage_sort),
           "--prelaunch", str(args.prelaunch), "--bind", args.bind]
    if args.two_streams:
        cmd.append("--two-streams")
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
        lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
        if lines:
            try:
                return json.loads(lines[-1])
            except ValueError:
                pass
        return {"extra_legs_error": f"child exited with {r.returncode}: {r.stderr.decode()[-300:]}"}
    except Exception as e:  # timeout, spawn failure
        return {"extra_legs_error": repr(e)[:300]}


def extra_legs(args):
    """The child: SURVEY 8(f) timings beside the headline.  Prints one JSON dict."""
    import torch

    bind_to_gpu(0, args.bind)

    M, N, sensor = CONFIGS[args.config]
    ext = bool(args.extrinsic_est)
    scene = synth.make_scene(M, synth.CONFIG_SEED_BASE + args.config)
    S = 3
    probs = [synth.make_problem(M, N, sensor, cfg=args.config, scan_seed=s, scene=scene) for s in range(S)]
    priors = [synth.propagate_prior_cov(capi.predict_fn, p.x_prior) for p in probs]
    priors = [(np.ascontiguousarray(x, np.float64), np.ascontiguousarray(P, np.float64)) for x, P in priors]
    bodies = []
    for p in probs:
        a = capi.pinned_empty((N, 3), np.float32)
        a[:] = p.body
        bodies.append(a)
    h = capi.Handle(cell_size=args.cell, lanes_per_query=args.lpq, sort_queries=args.sort, pass_kernel=args.pass_kernel,
                    index_cache=args.index_cache, stage_sort=args.stage_sort, prelaunch=args.prelaunch)
    h.map_build(scene.map_xyz)
    h.set_timing_stride(0)
    for s in range(S):
        h.scan_stage(16 + s, bodies[s])
    kf = capi.Esekf(h, max_iter=3, extrinsic_est_en=ext)
    out = {}

    if args.two_streams:
        # two independent scan streams in flight on this GPU (two handles, two host threads): while one stream's host solves
        # its 23x23 system the other's kernels run
        import threading

        h2 = capi.Handle(cell_size=args.cell, lanes_per_query=args.lpq, sort_queries=args.sort, pass_kernel=args.pass_kernel,
                          index_cache=args.index_cache, stage_sort=args.stage_sort, prelaunch=args.prelaunch)
        h2.map_build(scene.map_xyz)
        h2.set_timing_stride(0)
        for s in range(S):
            h2.scan_stage(16 + s, bodies[s])
        kf2 = capi.Esekf(h2, max_iter=3, extrinsic_est_en=ext)
        per = max(args.steps // 2, 8)
        jobs2 = capi.Esekf.make_jobs(bodies, priors, slots=[16 + s for s in range(S)])

        def worker(kfx, off):
            kfx.run_scans(jobs2, off, per, ring=RING)

        for kfx in (kf, kf2):
            worker(kfx, 0)
        torch.cuda.synchronize()
        th = [threading.Thread(target=worker, args=(kf, 0)), threading.Thread(target=worker, args=(kf2, 1))]
        t1 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        out["two_streams_per_gpu"] = {"scans_per_s": round(2 * per / dt2, 3), "scans": 2 * per,
                                      "note": "two independent scan streams (device-resident scans), one handle + one host thread each, same GPU"}
        kf2.close()
        h2.close()

    # ---- map_incremental (classification + Add_Points into the device map) after an update.
    # t_map = classify + insert + re-index, the reference's "Incremental Mapping" timer (src/laserMapping.cpp:921-924).
    t_cls = t_all = t_enq = 0.0
    added = 0
    for s in range(S):
        kf.update_scan(16 + s, priors[s][0], priors[s][1], 0.001)
        xpost = kf.get_x()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        h.map_incremental(xpost, 0.5, True, apply=False)
        t2 = time.perf_counter()
        m0 = h.M
        h.map_incremental(xpost, 0.5, True, apply=True, counts=False)  # as the node's loop calls it: the two counts are not asked for
        t3a = time.perf_counter()
        m1 = h.M  # the change is enqueued; reading the map's size waits for its counters (= the device has finished it)
        t3 = time.perf_counter()
        t_cls += t2 - t1
        t_all += t3 - t2
        t_enq += t3a - t2
        added += m1 - m0
    out["map_incremental"] = {"ms_per_scan": round(t_all / S * 1e3, 3), "classify_only_ms": round(t_cls / S * 1e3, 3),
                              "host_time_of_the_call_ms": round(t_enq / S * 1e3, 3),
                              "net_points_added_per_scan": round(added / S, 1), "scans": S, "changes": h.map_change_stats(),
                              "note": "filter_size_map 0.5; only the touched bricks are rewritten (slack-carrying brick storage); ms_per_scan = "
                                      "call + wait until the device has finished the change; the call itself only enqueues it"}

    # ---- the raw-scan front end (undistortion + VoxelGrid + staging) for one raw scan handed over
    # as a host buffer, and publish_frame_world's dense cloud: PCIe-inclusive by nature.
    rng = np.random.default_rng(11)
    body = probs[0].body
    raw = np.repeat(body, 3, axis=0) + rng.normal(0, 0.03, (3 * len(body), 3)).astype(np.float32)
    tms = np.repeat(rng.uniform(0.0, 100.0, len(body)), 3).astype(np.float32)  # neighbours in space are neighbours in time
    pts = capi.pinned_empty((len(raw), 4), np.float32)
    pts[:, :3] = raw
    pts[:, 3] = tms
    poses, x_end = synth.imu_poses(priors[0][0], capi.predict_fn)
    h.scan_stage_undistorted(0, pts, poses, x_end, 0.5, want_undistorted=False)  # warm-up (allocations)
    torch.cuda.synchronize()
    reps = 5
    t1 = time.perf_counter()
    for _ in range(reps):
        n_down, _u = h.scan_stage_undistorted(0, pts, poses, x_end, 0.5, want_undistorted=False)
    h.scan_wait(0)
    torch.cuda.synchronize()
    t_fe = (time.perf_counter() - t1) / reps
    t1 = time.perf_counter()
    for _ in range(reps):
        h.scan_stage_undistorted(0, pts, poses, x_end, 0.5, want_undistorted=True)
    t_fe_back = (time.perf_counter() - t1) / reps
    h.frame_world(x_end, slot=0, dense=True)
    t1 = time.perf_counter()
    for _ in range(reps):
        h.frame_world(x_end, slot=0, dense=True)
    t_fw = (time.perf_counter() - t1) / reps
    out["scan_front_end"] = {"raw_points": int(len(pts)), "feats_down_size": int(n_down),
                             "undistort_voxelgrid_stage_ms": round(t_fe * 1e3, 3),
                             "same_with_feats_undistort_copied_back_ms": round(t_fe_back * 1e3, 3),
                             "frame_world_dense_ms": round(t_fw * 1e3, 3),
                             "note": "page-locked host buffer in; filter_size_surf 0.5; frame_world = RGBpointBodyToWorld over "
                                     "the device-resident feats_undistort + D2H (publish_frame_world, dense_pub_en)"}
    print(json.dumps(out), flush=True)
    kf.close()
    h.close()


if __name__ == "__main__":
    main()
