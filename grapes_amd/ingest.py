"""Graph / feature ingest for the partitioned path: `edge_index` -> CSR with the SciPy constructor's
semantics (main.py:134-136: duplicates collapse, columns ascending) is `graph.DeviceGraph.from_edge_index`; this
module cuts a CSR + feature matrix into the 1-D node-range shards of `dist.PartitionedGraph` and keeps them in an
on-disk cache, so that an 8-GPU job on a papers100M-sized graph never builds the full CSR in one process: every rank
loads only its own shard (rowptr rebased to 0, GLOBAL column ids, its feature rows).

Cache layout (one directory per (graph, world size)):
    meta.json                       {"num_nodes", "world", "bounds", "feature_dim", "nnz", "version"}
    shard_<rank>.npz                rowptr int64[n_loc+1], col int32[nnz_loc], x float32[n_loc, F], y (optional)
Plain .npz (numpy) — no pickled code, readable without torch.
"""
from __future__ import annotations

import json
import os
from typing import Optional

import numpy as np
import torch

from .dist import PartitionedGraph, partition_bounds

CACHE_VERSION = 1


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def write_partition_cache(path: str, rowptr, col, X, world: int, y=None) -> dict:
    """Cuts (rowptr int64[N+1], col int32/int64[nnz], X fp32[N,F]) into `world` contiguous node ranges and writes one
    shard file per rank.  Inputs may be numpy arrays or (CPU / device) tensors; returns the meta record."""
    rowptr, col, X = _np(rowptr).astype(np.int64), _np(col).astype(np.int32), _np(X).astype(np.float32)
    N = rowptr.shape[0] - 1
    if X.shape[0] != N:
        raise ValueError("feature matrix and CSR disagree on the number of nodes")
    if N >= 2 ** 31:
        raise ValueError("node ids must fit int32")
    bounds = partition_bounds(N, world)
    os.makedirs(path, exist_ok=True)
    for r in range(world):
        lo, hi = bounds[r], bounds[r + 1]
        a, b = int(rowptr[lo]), int(rowptr[hi])
        arrays = dict(rowptr=rowptr[lo:hi + 1] - rowptr[lo], col=col[a:b], x=X[lo:hi])
        if y is not None:
            arrays["y"] = _np(y)[lo:hi]
        np.savez(os.path.join(path, f"shard_{r}.npz"), **arrays)
    meta = dict(num_nodes=int(N), world=int(world), bounds=[int(v) for v in bounds], feature_dim=int(X.shape[1]),
                nnz=int(rowptr[-1]), version=CACHE_VERSION)
    with open(os.path.join(path, "meta.json"), "w") as f:
        json.dump(meta, f)
    return meta


def read_meta(path: str) -> dict:
    with open(os.path.join(path, "meta.json")) as f:
        meta = json.load(f)
    if meta.get("version") != CACHE_VERSION:
        raise ValueError(f"partition cache {path}: unsupported version {meta.get('version')}")
    return meta


def load_partition(path: str, rank: int, world: int, device="cuda", group=None, local_ops=None,
                   slot_factor: float = 2.0, return_labels: bool = False):
    """This rank's shard as a dist.PartitionedGraph (and its label slice if stored and requested)."""
    meta = read_meta(path)
    if meta["world"] != world:
        raise ValueError(f"partition cache {path} was written for world size {meta['world']}, not {world}")
    z = np.load(os.path.join(path, f"shard_{rank}.npz"))
    rp = torch.from_numpy(z["rowptr"]).to(device)
    cl = torch.from_numpy(z["col"]).to(device)
    x = torch.from_numpy(z["x"]).to(device)
    max_degree = int((rp[1:] - rp[:-1]).max().item()) if rp.numel() > 1 else 0
    g = PartitionedGraph(rp, cl, x, meta["bounds"], rank, world, group, local_ops, max_degree, slot_factor=slot_factor)
    if return_labels:
        return g, (torch.from_numpy(z["y"]).to(device) if "y" in z.files else None)
    return g
