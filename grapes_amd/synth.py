"""Seeded synthetic graphs with the statistics of the benchmark datasets: no dataset
files or network exist on either box.  Node weights w = min(pareto(2.2)+1, mean(w)*maxdeg/avgdeg),
endpoints drawn proportionally to w, symmetrised, self-loops dropped, deduplicated by the CSR build.

Two generators with the same recipe: numpy (CPU tests, small graphs) and torch-on-device (bench:
products scale, 1.2e8 directed edges, built in HBM in a few seconds).  They are not bit-identical to
each other; parity tests always feed the SAME arrays to the oracle and to the HIP path.
"""
from __future__ import annotations

import math

import numpy as np
import torch

# name -> (N, avg_deg, max_deg, F, C, B, K, hops)
CONFIGS = {
    "cora": (2708, 3.9, 168, 1433, 7, 512, 16, 2),
    "arxiv": (169343, 13.7, 13161, 128, 40, 256, 256, 2),
    "reddit": (232965, 99.6, 21657, 602, 41, 256, 512, 2),
    "products": (2449029, 50.5, 17481, 100, 47, 256, 256, 3),
    # BASELINE config 5 (SURVEY §8d "papers100M-like"; the reference has no loader, modules/data.py:283-292): the symmetrised
    # graph, ~3.2e9 directed edges, F=128, C=172, 3 hops.  Built by synth_graph_device_chunked (64-bit offsets throughout).
    "papers100m": (111059956, 29.0, 30000, 128, 172, 256, 256, 3),
}


def _weights_np(rng, n, avg_deg, max_deg):
    w = rng.pareto(2.2, n) + 1.0
    return np.minimum(w, w.mean() * max_deg / avg_deg)


def synth_edges_numpy(n, avg_deg, max_deg, seed=0):
    rng = np.random.default_rng(seed)
    w = _weights_np(rng, n, avg_deg, max_deg)
    cdf = np.cumsum(w)
    m = int(n * avg_deg / 2)
    a = np.searchsorted(cdf, rng.random(m) * cdf[-1]).clip(0, n - 1)
    b = np.searchsorted(cdf, rng.random(m) * cdf[-1]).clip(0, n - 1)
    keep = a != b
    a, b = a[keep], b[keep]
    return np.stack([np.concatenate([a, b]), np.concatenate([b, a])]).astype(np.int64)


def synth_csr_numpy(n, avg_deg, max_deg, seed=0):
    ei = synth_edges_numpy(n, avg_deg, max_deg, seed)
    key = np.unique(ei[0] * np.int64(n) + ei[1])
    r, c = key // n, (key % n).astype(np.int32)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(indptr, r + 1, 1)
    return np.cumsum(indptr), c


def synth_graph_device(n, avg_deg, max_deg, seed=0, device="cuda"):
    """(rowptr int64[N+1], col int32[nnz]) built on the device with torch ops (ingest plumbing)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    u = torch.rand(n, device=device, generator=gen, dtype=torch.float64).clamp_(1e-12, 1.0)
    w = u.pow(-1.0 / 2.2)                                  # pareto(2.2) + 1
    w = torch.minimum(w, w.mean() * (max_deg / avg_deg))
    cdf = torch.cumsum(w, 0)
    m = int(n * avg_deg / 2)
    tot = cdf[-1]
    a = torch.searchsorted(cdf, torch.rand(m, device=device, generator=gen, dtype=torch.float64) * tot).clamp_(0, n - 1)
    b = torch.searchsorted(cdf, torch.rand(m, device=device, generator=gen, dtype=torch.float64) * tot).clamp_(0, n - 1)
    keep = a != b
    a, b = a[keep], b[keep]
    key = torch.cat([a * n + b, b * n + a])
    del a, b, keep, cdf, w, u
    key = torch.unique(key)
    row = torch.div(key, n, rounding_mode="floor")
    col = (key - row * n).to(torch.int32)
    del key
    counts = torch.bincount(row, minlength=n)
    del row
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    return rowptr, col


def synth_graph_device_chunked(n, avg_deg, max_deg, seed=0, device="cuda", chunk=1 << 27, nnz_scale=1.0):
    """The same recipe for graphs whose edge list does not fit a torch sort (papers100M: 3.2e9 directed edges): endpoints are
    drawn in chunks straight into one int64 edge_index [2, 2m] (51 GB at papers100M size) and the CSR comes from the library's
    own ingest kernel (ops.csr_build: counting placement + per-row sort / de-duplication, SciPy constructor semantics) instead
    of torch.unique over 64-bit keys.  nnz_scale < 1 thins the graph (tests).  Self-pairs are redirected to the next node
    instead of dropped (the edge array has a fixed size); the CSR keeps a self-loop out either way."""
    from . import ops
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    u = torch.rand(n, device=device, generator=gen, dtype=torch.float64).clamp_(1e-12, 1.0)
    w = u.pow_(-1.0 / 2.2)                                 # pareto(2.2) + 1
    w = torch.minimum(w, w.mean() * (max_deg / avg_deg))
    cdf = torch.cumsum(w, 0)
    del w, u
    tot = cdf[-1]
    m = int(n * avg_deg * nnz_scale / 2)
    ei = torch.empty((2, 2 * m), dtype=torch.int64, device=device)
    for lo in range(0, m, chunk):
        c = min(chunk, m - lo)
        a = torch.searchsorted(cdf, torch.rand(c, device=device, generator=gen, dtype=torch.float64) * tot).clamp_(0, n - 1)
        b = torch.searchsorted(cdf, torch.rand(c, device=device, generator=gen, dtype=torch.float64) * tot).clamp_(0, n - 1)
        b = torch.where(a == b, (b + 1) % n, b)
        ei[0, lo:lo + c] = a; ei[1, lo:lo + c] = b
        ei[0, m + lo:m + lo + c] = b; ei[1, m + lo:m + lo + c] = a
        del a, b
    del cdf
    rowptr, col = ops.csr_build(ei, n)
    del ei
    torch.cuda.empty_cache()
    return rowptr, col


def randn_rows_(X: torch.Tensor, generator=None, rows_per_chunk=1 << 22):
    """X ~ N(0, 1) filled in row chunks (a 57 GB matrix in one normal_ call would need 64-bit element counters)."""
    for lo in range(0, X.shape[0], rows_per_chunk):
        X[lo:lo + rows_per_chunk].normal_(generator=generator)
    return X


def glorot_(weight: torch.Tensor):
    fo, fi = weight.shape
    a = math.sqrt(6.0 / (fi + fo))
    with torch.no_grad():
        weight.uniform_(-a, a)
