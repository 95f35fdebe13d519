"""Partition cache (grapes_amd/ingest.py) on the CPU: shards written from a full CSR reproduce
dist.shard_full_graph for every rank."""
import numpy as np
import pytest
import torch

from grapes_amd import ingest
from grapes_amd.dist import partition_bounds, shard_full_graph
from oracle import grapes_oracle as O


@pytest.mark.parametrize("world", [1, 3, 8])
def test_partition_cache_roundtrip(tmp_path, world):
    rng = np.random.default_rng(world)
    N, F = 1003, 5
    ei = rng.integers(0, N, (2, 7000))
    indptr, indices = O.build_csr(np.concatenate([ei, ei[::-1]], axis=1), N)          # main.py:134-136 semantics
    X = rng.standard_normal((N, F)).astype(np.float32)
    y = rng.integers(0, 7, N)
    meta = ingest.write_partition_cache(str(tmp_path), indptr, indices, X, world, y=y)
    assert meta["bounds"] == partition_bounds(N, world) and meta["nnz"] == int(indptr[-1])
    for rank in range(world):
        g, yl = ingest.load_partition(str(tmp_path), rank, world, device="cpu", return_labels=True)
        ref = shard_full_graph(torch.from_numpy(indptr), torch.from_numpy(indices.astype(np.int32)), torch.from_numpy(X), rank, world)
        assert g.lo == ref.lo and g.hi == ref.hi and g.bounds == ref.bounds
        assert torch.equal(g.rowptr, ref.rowptr) and torch.equal(g.col, ref.col) and torch.equal(g.X, ref.X)
        assert np.array_equal(yl.numpy(), y[g.lo:g.hi])
    with pytest.raises(ValueError):
        ingest.load_partition(str(tmp_path), 0, world + 1, device="cpu")
