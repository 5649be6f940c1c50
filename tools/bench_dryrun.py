"""CPU dry run of bench.py: the device layer (capi.Handle, the filter bound to a handle, page-locked allocation, the torch.cuda
calls) is replaced by fakes that return fixed numbers, everything else -- argument handling, scene and scan generation, the
legs and their bookkeeping, the torch.distributed calls (gloo), the CPU baseline, the JSON line -- is bench.py's own code.
It exists to catch Python-level mistakes in bench.py on a machine without a GPU (tests/test_bench_dryrun.py); the numbers it
prints mean nothing.

  python tools/bench_dryrun.py --leg main --config 1 --steps 6 --warmup 2 --scans 3 --cpu-scans 1 [--force-shard-leg]
  python -m torch.distributed.run --nproc-per-node 2 ... tools/bench_dryrun.py --gpus 2 --backend gloo --single-device 1 ...
"""
import sys, types, json
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fast_lio_amd import capi
import importlib.util

class FakeHandle:
    def __init__(self, **kw):
        self.kw = kw; self.M = 0; self._n = 0; self.stride = 1
    def map_build(self, xyz): self.M = len(xyz)
    def set_timing_stride(self, n): self.stride = n
    def set_timing_sampling(self, n, search_only): self.stride = n
    def counters(self, reset=False):
        return {"search_ms": 0.5, "n_search": 10, "fit_ms": 0.3, "n_fit": 20, "eval_ms": 1.0, "n_eval": 20}
    def search_counters(self):
        return {"first_ms": 0.3, "n_first": 5, "later_ms": 0.2, "n_later": 5}
    def scan_stage(self, slot, body): assert 0 <= slot < 64
    def scan_upload(self, body): pass
    def enable_stats(self, on): pass
    def eval(self, x, search, ext): return np.zeros((12, 12)), np.zeros(12), 100, 1.0
    def timing(self): return {"search_ms": 0.04, "fit_ms": 0.01, "total_ms": 0.05, "candidates": 8_000_000}
    def close(self): pass
    def rccl_init_rank(self, n, uid, r): assert len(uid) == 128
    def rccl_size(self): return 1
    def peer_open(self, name, n, r):
        assert name.startswith("/")
        if os.environ.get("DRYRUN_PEER_FAILS_ON_RANK") == str(r):  # tests/test_bench_dryrun.py: the fallback to the other exchange
            raise RuntimeError("flh_peer_open: hipHostRegister: out of memory (dry-run fault injection)")
    def peer_size(self): return 1
    def debug_bounds(self): return False, [0] * 20
    def pass_stats(self): return {"search_passes": 40, "one_launch_passes": 40, "second_stage_queries": 4000, "nosearch_passes": 40}
    def prelaunch_stats(self): return {"armed": 80, "go": 78, "abort": 2, "gone": 0}
    def set_owned_interval(self, a, lo, hi): pass
    def map_incremental(self, x, fsm, inited, apply=True, counts=True): self.M += 10 if apply else 0; return (5, 5) if counts else None
    def map_change_stats(self): return {"enqueued_without_wait": 2, "replayed": 0}
    def scan_stage_undistorted(self, slot, pts, poses, x_end, leaf, want_undistorted=True): return 1234, None
    def scan_wait(self, slot): pass
    def frame_world(self, x, slot=-1, dense=True): return np.zeros((10, 3), np.float32)

class FakeRS:
    passes = 80; searches = 40; ms_search_passes = 2.4; n_search_passes = 40; ms_nosearch_passes = 0.9; n_nosearch_passes = 40
    ms_map_incremental = 1.0
RealEsekf = capi.Esekf
class FakeEsekf:
    make_jobs = staticmethod(capi.Esekf.make_jobs)
    def __new__(cls, h=None, *a, **k):
        if h is None:
            return RealEsekf(None)
        return object.__new__(cls)
    def __init__(self, h, max_iter=3, extrinsic_est_en=False): self.h = h
    def run_scans(self, jobs, first, count, ring=4, R=0.001, map_incremental=False, filter_size_map=0.5, first_staged=False, stage_next=False):
        assert len(jobs) >= 1 and count >= 0
        return FakeRS()
    def update_scan(self, slot, x, P, R=0.001): pass
    def get_x(self): return np.zeros(26)
    def close(self): pass

real_predict = capi.predict_fn
capi.Handle = FakeHandle
capi.Esekf = FakeEsekf
capi.device_available = lambda: True
capi.pinned_empty = lambda shape, dtype=np.float32: np.empty(shape, dtype)
capi.rccl_unique_id = lambda: b"\0" * 128
torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a: None
_tt = torch.tensor
torch.tensor = lambda *a, **k: _tt(*a, **{kk: vv for kk, vv in k.items() if kk != 'device'})

spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py')); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
b.run_extra_legs_in_child = lambda args: {"extras": "skipped in the dry run"}
sys.argv = ['bench.py'] + sys.argv[1:]
b.main()
