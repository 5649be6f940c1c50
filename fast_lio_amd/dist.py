"""Multi-GPU glue for the measurement update (stage C1).

One process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests).  The scan's
points are sharded contiguously over the ranks, the map is replicated; every h_share_model evaluation each
rank reduces its shard to the 16x16 Gram block G = sum_k v_k v_k^T (v = [row(12) | h | 1 | |pd2| | 0]) and
ONE all-reduce of those 256 doubles (2 KB, latency-bound, nothing else crosses xGMI) gives every rank the
same normal equations; the 23x23 solve then runs identically on every rank.

The functions here are transport-agnostic: `allreduce` is any callable that sums a 256-vector in place
across ranks, `eval_partial` any callable producing the local Gram block.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

NDOF = capi.NDOF


def shard_bounds(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced shard [lo, hi) of n scan points for `rank` of `world`."""
    return (rank * n) // world, ((rank + 1) * n) // world


def pack_gram(HTH: np.ndarray, HTh: np.ndarray, n_eff: int, total_residual: float) -> np.ndarray:
    """Inverse of flh_unpack_gram: the 256-double layout the device reduction produces."""
    G = np.zeros((16, 16))
    G[:12, :12] = np.asarray(HTH, np.float64).reshape(12, 12)
    G[:12, 12] = HTh
    G[12, :12] = HTh
    G[13, 13] = float(n_eff)
    G[14, 13] = float(total_residual)
    return G.reshape(256)


def unpack_gram(G: np.ndarray):
    HTH = np.zeros(144)
    HTh = np.zeros(12)
    n = C.c_int64()
    tr = C.c_double()
    capi.lib().flh_unpack_gram(np.ascontiguousarray(G, np.float64).reshape(256), HTH, HTh, C.byref(n), C.byref(tr))
    return HTH.reshape(12, 12), HTh, int(n.value), float(tr.value)


def make_sharded_model(eval_partial, allreduce, gather_rows=None):
    """Measurement model for Esekf.set_meas_model that sums the ranks' partial normal equations.

    eval_partial(x, converge) -> object holding the local 256-double Gram block (numpy array or torch tensor)
    allreduce(obj)            -> sums it in place across ranks and returns a numpy view/copy of the result
    gather_rows(x)            -> (h_x (n x 12), h (n)) of ALL ranks in scan order; only called when the global
                                 n_eff < 23 (gain-form branch, esekfom.hpp:1715-1744)
    """

    def model(x, converge):
        g = eval_partial(x, converge)
        G = allreduce(g)
        HTH, HTh, n_eff, tres = unpack_gram(G)
        if n_eff < 1:
            return {"valid": False, "n_eff": 0}
        out = {"valid": True, "n_eff": n_eff, "HTH": HTH, "HTh": HTh, "total_residual": tres}
        if n_eff < NDOF:
            if gather_rows is None:
                raise RuntimeError("sharded update: fewer than 23 effective points and no gather_rows callback")
            hx, hv = gather_rows(x)
            out["h_x"], out["h"] = hx, hv
        return out

    return model


def torch_allreduce(dist_module, tensor):
    """all-reduce (sum) a torch tensor holding the Gram block; returns it as a numpy array on the host."""
    dist_module.all_reduce(tensor)
    return tensor.detach().cpu().numpy()
