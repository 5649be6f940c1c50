"""bench.py's own Python (legs, bookkeeping, torch.distributed calls, the JSON contract) run on the CPU with the device layer
faked (tools/bench_dryrun.py): a typo in a leg must not wait for a GPU box to be found."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRY = os.path.join(ROOT, "tools", "bench_dryrun.py")
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def _line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_one_rank_all_legs_of_the_main_process():
    r = subprocess.run([sys.executable, DRY, "--leg", "main", "--config", "1", "--steps", "6", "--warmup", "2", "--scans", "3",
                        "--cpu-scans", "1", "--force-shard-leg"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    d = _line(r.stdout.decode())
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["config"]["workload"].startswith("BASELINE configs[0]")
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 3
    if (os.cpu_count() or 1) >= 8:  # the best of a small thread sweep, not "all cores" (the restated path anti-scales there)
        assert d["cpu_baseline"]["best_of_thread_sweep"]["cores"] in (8, 16, 32, 64)
    assert d["config"]["distinct_scans"] == 3 and set(d["roofline"]["events_sampled_by_kind"]) == {"first_search_of_scan", "later_search"}
    assert d["shard_mode"]["ranks_in_communicator"] == 1 and "error" not in d["shard_mode"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"]


def test_partition_mode_and_the_side_legs():
    r = subprocess.run([sys.executable, DRY, "--leg", "main", "--config", "1", "--steps", "6", "--warmup", "2", "--scans", "3",
                        "--cpu-scans", "0", "--force-shard-leg", "--mode", "partition"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    d = _line(r.stdout.decode())
    assert d["scaling"] == "strong"
    r = subprocess.run([sys.executable, DRY, "--leg", "extras", "--config", "1", "--steps", "8", "--two-streams"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    d = _line(r.stdout.decode())
    assert set(d) >= {"two_streams_per_gpu", "map_incremental", "scan_front_end"}


def test_two_ranks_over_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", DRY, "--gpus", "2", "--config", "1", "--steps", "6", "--warmup", "2", "--scans", "3",
                        "--cpu-scans", "0", "--backend", "gloo", "--single-device", "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900, env=env)
    d = _line(r.stdout.decode())
    # N > 1: the headline is the north_star split (ONE scan sharded over the ranks + all-reduce inside every pass), strong
    # scaling; the N independent replicas are a labelled sub-field and never `value`
    assert d["n_gpus"] == 2 and d["ranks_seen_by_collective"] == 2 and d["scaling"] == "strong"
    assert "sharded" in d["config"]["parallelism"] and d["ranks_exchanging_normal_equations"] is not None
    # the default exchange: peer-written granules (no collective); the RCCL all-reduce is timed beside it
    assert d["sharded_path"]["collective"].startswith("peer granules") and "shard_mode" not in d
    assert d["other_exchange"]["collective"].startswith("rccl") and "error" not in d["other_exchange"]
    assert d["replicas_no_collective"]["scaling"] == "weak" and "NO collective" in d["replicas_no_collective"]["note"]
    assert abs(d["value"] * d["ms_per_step"] / 1000.0 - 1.0) < 0.01  # value = scans of the ONE sharded stream per second (not x ranks)


def test_two_ranks_fall_back_to_the_other_exchange_when_one_rank_cannot_open_the_peers():
    """The sharded leg is the headline for N > 1: when the chosen exchange does not come up on EVERY rank (here: rank 1 cannot
    page-lock the shared segment), all ranks agree to measure with the other one and the line says so."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", DRYRUN_PEER_FAILS_ON_RANK="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29543", DRY, "--gpus", "2", "--config", "1", "--steps", "6", "--warmup", "2", "--scans", "3",
                        "--cpu-scans", "0", "--backend", "gloo", "--single-device", "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900, env=env)
    d = _line(r.stdout.decode())
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["sharded_path"]["collective"].startswith("rccl") and "flh_peer_open" in d["sharded_path"]["exchange_fallback"]
    assert "other_exchange" not in d  # the other exchange IS the headline now; nothing is timed beside it
    assert abs(d["value"] * d["ms_per_step"] / 1000.0 - 1.0) < 0.01


def test_without_a_device_the_real_bench_says_so_and_does_not_retry():
    from fast_lio_amd import capi

    if capi.device_available():
        import pytest

        pytest.skip("a device is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--scans", "2", "--config", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    err = r.stderr.decode()
    assert r.returncode == 3 and "needs a GPU" in err and "attempt 2/2" not in err and r.stdout.decode().strip() == ""
