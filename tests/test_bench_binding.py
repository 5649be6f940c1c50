"""bench.py's host binding: which CPUs the process of a GPU gets (pure logic; the topology of the two-socket MI355X box of
profiles/r06_call37/: 2 x 64 cores, SMT siblings at +128, 8-core CCDs, four GPUs per NUMA node)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("bench_for_binding_test", os.path.join(ROOT, "bench.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    finally:
        sys.argv = argv


def _box():
    local = list(range(0, 64)) + list(range(128, 192))  # node 0
    sib = {c: [c % 128, c % 128 + 128] for c in local}
    l3_of = lambda c: [x for x in local if (x % 128) // 8 == (c % 128) // 8]  # noqa: E731
    gpus = ["0000:0a:00.0", "0000:23:00.0", "0000:5a:00.0", "0000:72:00.0"]
    return local, sib, l3_of, gpus


def test_cpulist_parser():
    b = _bench()
    assert b._cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert b._cpulist("") == []


def test_modes_on_the_two_socket_box():
    b = _bench()
    local, sib, l3_of, gpus = _box()
    bdf = "0000:5a:00.0"  # third of the node's four GPUs
    cpus, note = b.binding_cpus("l3", bdf, 0, local, gpus, sib, l3_of)
    assert cpus == list(range(32, 40)) and "behind one L3" in note and "SMT" not in note
    cpus, _ = b.binding_cpus("l3smt", bdf, 0, local, gpus, sib, l3_of)
    assert cpus == list(range(32, 40)) + list(range(160, 168))
    cpus, _ = b.binding_cpus("share", bdf, 0, local, gpus, sib, l3_of)
    assert cpus == list(range(32, 48)) + list(range(160, 176))
    cpus, _ = b.binding_cpus("node", bdf, 0, local, gpus, sib, l3_of)
    assert cpus == sorted(local)


def test_the_gpus_of_a_node_get_disjoint_cores():
    b = _bench()
    local, sib, l3_of, gpus = _box()
    seen = set()
    for g in gpus:
        cpus, _ = b.binding_cpus("l3", g, 0, local, gpus, sib, l3_of)
        assert len(cpus) == 8 and not (seen & set(cpus))
        seen |= set(cpus)


def test_fallbacks():
    b = _bench()
    local, sib, l3_of, gpus = _box()
    # a GPU that is not in the node's list, or too few cores to divide: the whole node
    cpus, note = b.binding_cpus("l3", "0000:ff:00.0", 0, local, gpus, sib, l3_of)
    assert cpus == sorted(local) and note.startswith("NUMA node")
    few = list(range(0, 8))
    cpus, _ = b.binding_cpus("l3", gpus[0], 0, few, gpus, {c: [c] for c in few}, lambda c: few)
    assert cpus == few
    # a restricted affinity mask (cgroup): siblings outside it are not used
    half = list(range(0, 64))
    cpus, _ = b.binding_cpus("share", gpus[0], 0, half, gpus, {c: [c, c + 128] for c in half}, l3_of)
    assert cpus == list(range(0, 16))
    # an L3 group smaller than four cores of the share: the share stays whole
    cpus, _ = b.binding_cpus("l3", gpus[0], 0, local, gpus, sib, lambda c: [c, c + 128])
    assert cpus == list(range(0, 16))
