"""Host logic of peer.PeerFeatures (the shard table the fused gather-SpMM picks rows from): bounds, padding, table layout and the
refusals — no GPU, no compute call (in-process shards over CPU tensors only exercise the bookkeeping)."""
import ctypes as C

import pytest
import torch

from grapes_amd.dist import partition_bounds
from grapes_amd.peer import MAX_SHARDS, PeerFeatures


def test_table_of_in_process_shards():
    X = torch.arange(50 * 6, dtype=torch.float32).reshape(50, 6)          # F = 6 -> rows padded to 8 floats
    cuts = [0, 7, 7, 30, 50]                                              # one empty shard
    shards = [X[a:b].clone() for a, b in zip(cuts, cuts[1:])]
    pf = PeerFeatures.from_shards(shards, rank=2)
    assert pf.P == 4 and pf.bounds == cuts and pf.F == 6 and pf.pitch == 8 and pf.shape == (50, 6) and pf.rank == 2
    bases, bounds, P = pf.c_table()
    assert P == 4 and list(bounds) == cuts and len(bases) == 4 and all(b for b in bases)        # no NULLs, even for the empty shard
    assert pf.local.shape == (23, 8) and torch.equal(pf.local[:, :6], X[7:30]) and float(pf.local[:, 6:].abs().max()) == 0.0
    assert pf.contiguous() is pf and pf.is_cuda                          # what step_graph.GraphedTrainer asks of X


def test_refusals():
    X = torch.zeros(4, 4)
    with pytest.raises(ValueError):
        PeerFeatures.from_shards([X] * (MAX_SHARDS + 1))                  # one node: at most 8 shards
    with pytest.raises(ValueError):
        PeerFeatures(X, 4, [1, 4], 0, [X.data_ptr()])                     # bounds must start at 0
    with pytest.raises(ValueError):
        PeerFeatures(X, 4, [0, 3, 2], 0, [X.data_ptr(), X.data_ptr()])    # ... and ascend


def test_partition_bounds_cover_the_nodes():
    for n, p in ((2449029, 8), (111059956, 8), (10, 3), (5, 8)):
        b = partition_bounds(n, p)
        assert b[0] == 0 and b[-1] == n and len(b) == p + 1 and all(y >= x for x, y in zip(b, b[1:]))
        assert max(y - x for x, y in zip(b, b[1:])) - min(y - x for x, y in zip(b, b[1:])) <= 1
