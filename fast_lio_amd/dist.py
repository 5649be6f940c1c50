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


def morton_order(body: np.ndarray, quantum: float = 0.25) -> np.ndarray:
    """Permutation that sorts scan points by the Morton code of their body-frame coordinates (the order the device keeps a
    staged scan in).  Sharding a scan Morton-first (shard = perm[lo:hi]) gives every rank a spatially compact piece, so
    its queries walk a compact part of the map."""
    q = np.clip(np.asarray(body, np.float32) / np.float32(quantum) + np.float32(8192.0), 0, 16383).astype(np.uint64)

    def spread(v):
        v = v & np.uint64(0x3FFF)
        v = (v | (v << np.uint64(16))) & np.uint64(0x0000FF0000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x00F00F00F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x0C30C30C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x249249249249)
        return v

    key = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))
    return np.argsort(key, kind="stable")


def morton_shard(body: np.ndarray, rank: int, world: int) -> np.ndarray:
    """Indices (into the scan, ascending) of this rank's Morton-first shard."""
    lo, hi = shard_bounds(len(body), rank, world)
    return np.sort(morton_order(body)[lo:hi])


# ---------------------------------------------------------------- map partitioned over the ranks (BASELINE configs[4])
def partition_axis(map_xyz: np.ndarray) -> int:
    """The axis the map is cut along: the longest extent."""
    ext = np.ptp(np.asarray(map_xyz)[:, :3], axis=0)
    return int(np.argmax(ext))


def partition_bounds(map_xyz: np.ndarray, world: int, axis: int | None = None):
    """Cuts the map into `world` slabs of (nearly) equal point count along `axis`.  Returns (axis, edges) with
    edges[0] = -inf < edges[1] <= ... < edges[world] = +inf: rank r OWNS the queries whose world coordinate lies in
    [edges[r], edges[r+1]) -- the intervals tile the axis, so every query has exactly one owner."""
    if axis is None:
        axis = partition_axis(map_xyz)
    c = np.asarray(map_xyz, np.float32)[:, axis]
    qs = np.quantile(c, np.linspace(0, 1, world + 1)[1:-1]).astype(np.float32) if world > 1 else np.zeros(0, np.float32)
    edges = np.concatenate([[-np.inf], np.unique(qs) if world > 1 else [], [np.inf]]).astype(np.float32)
    if len(edges) != world + 1:
        raise ValueError("map too degenerate along the partition axis for this many ranks")
    return axis, edges


def partition_slab(map_xyz: np.ndarray, axis: int, edges: np.ndarray, rank: int, halo: float) -> np.ndarray:
    """Indices (ascending, so the tie-break order of the full map is kept) of the map points rank `rank` must hold: its slab
    plus a halo.  halo >= sqrt(max_sqdist) + a margin for the fp32 world coordinates: a query's five neighbours can only
    matter when they all lie within sqrt(max_sqdist) of it (src/laserMapping.cpp:671), and then they are all in here."""
    c = np.asarray(map_xyz, np.float32)[:, axis]
    lo, hi = float(edges[rank]) - halo, float(edges[rank + 1]) + halo
    return np.nonzero((c >= lo) & (c < hi))[0]


HALO_DEFAULT = float(np.sqrt(5.0)) + 0.05


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
