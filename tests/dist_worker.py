"""Worker for tests/test_dist_cpu.py: exercises grapes_amd.dist.PartitionedGraph over the gloo
backend on the CPU (world_size >= 2).  The local kernels are replaced by an oracle-backed test
double (tests may use the oracle; the product never does)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grapes_amd.dist import PartitionedGraph, make_grad_sync, partition_bounds, shard_full_graph  # noqa: E402
from oracle import grapes_oracle as O  # noqa: E402


class OracleLocalOps:
    def offsets(self, rowptr, nodes32):
        n = nodes32.long()
        lens = rowptr[n + 1] - rowptr[n]
        eoff = torch.zeros(n.numel() + 1, dtype=torch.int32)
        eoff[1:] = torch.cumsum(lens, 0).to(torch.int32)
        return eoff, eoff[-1:].clone()

    def expand(self, rowptr, col, nodes32, eoff, e_cap, want_pos=False):
        # like the HIP kernel, the output layout is dictated by eoff (a node whose slot has length 0 emits nothing)
        n = nodes32.long()
        lens = (eoff[1:] - eoff[:-1]).long()
        e = int(lens.sum())
        starts = rowptr[n]
        rep = torch.repeat_interleave(torch.arange(n.numel()), lens)
        within = torch.arange(e) - torch.repeat_interleave(eoff[:-1].long(), lens)
        src = torch.zeros(e_cap, dtype=torch.int32); dst = torch.zeros(e_cap, dtype=torch.int32)
        src[:e] = n[rep].to(torch.int32)
        dst[:e] = col[starts[rep] + within].to(torch.int32)
        pos = None
        if want_pos:
            pos = torch.zeros(e_cap, dtype=torch.int32)
            pos[:e] = rep.to(torch.int32)
        return src, dst, pos

    def gather_rows(self, X, ids32):
        return X[ids32.long()].contiguous()

    def take(self, table32, keys32):
        return table32[keys32.long()]


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(123)                      # same full graph on every rank
    N, F = 5003, 7
    ei = rng.integers(0, N, (2, 40000))
    ei[0, :3000] = 17                                     # a hub
    indptr, indices = O.build_csr(np.concatenate([ei, ei[::-1]], axis=1), N)
    X = torch.from_numpy(rng.standard_normal((N, F)).astype(np.float32))
    g = shard_full_graph(torch.from_numpy(indptr), torch.from_numpy(indices), X, rank, world,
                         local_ops=OracleLocalOps())
    b = partition_bounds(N, world)
    assert g.lo == b[rank] and g.hi == b[rank + 1] and b[0] == 0 and b[-1] == N
    qrng = np.random.default_rng(1000 + rank)             # a DIFFERENT query per rank (data-parallel batches)
    cases = [qrng.permutation(N)[:300], np.array([17, 5, 17, N - 1, 0]),
             np.arange(b[0], min(b[1], 40)),              # everything owned by rank 0
             qrng.permutation(N)[:1], np.zeros(0, np.int64)]
    for nodes in cases:
        nodes = np.asarray(nodes, dtype=np.int64)
        ref = O.get_neighborhoods(nodes, indptr, indices)
        e = ref.shape[1]
        src, dst, d_e = g.expand(torch.from_numpy(nodes.astype(np.int32)), e + 13)
        assert int(d_e.item()) == e, (rank, int(d_e.item()), e)
        assert np.array_equal(src[:e].numpy().astype(np.int64), ref[0]), rank      # query order preserved
        assert np.array_equal(dst[:e].numpy().astype(np.int64), ref[1]), rank      # ascending column inside a row
    # capacity-padded query: only the first *d_m entries count, the padding holds arbitrary (valid) ids
    nodes = qrng.permutation(N)[:200].astype(np.int64)
    m_true = 120 + 7 * rank
    ref = O.get_neighborhoods(nodes[:m_true], indptr, indices)
    src, dst, d_e = g.expand(torch.from_numpy(nodes.astype(np.int32)), ref.shape[1] + 5,
                             d_m=torch.tensor([m_true], dtype=torch.int32))
    e = ref.shape[1]
    assert int(d_e.item()) == e
    assert np.array_equal(src[:e].numpy().astype(np.int64), ref[0]) and np.array_equal(dst[:e].numpy().astype(np.int64), ref[1])
    ids = np.sort(qrng.permutation(N)[:300]).astype(np.int32)
    out = g.features(torch.from_numpy(ids), d_n=torch.tensor([211], dtype=torch.int32))
    assert out.shape == (211, F) and torch.equal(out, X[torch.from_numpy(ids[:211]).long()])
    for n_ids in (500, 1, 0):
        ids = np.sort(qrng.permutation(N)[:n_ids]).astype(np.int32)
        out = g.features(torch.from_numpy(ids))
        assert out.shape == (n_ids, F)
        assert torch.equal(out, X[torch.from_numpy(ids).long()]), rank             # halo rows are bit copies
    # gradient all-reduce (mean)
    p = torch.nn.Parameter(torch.zeros(5))
    p.grad = torch.full((5,), float(rank + 1))
    make_grad_sync(world)([p])
    assert torch.allclose(p.grad, torch.full((5,), sum(range(1, world + 1)) / world))
    assert g.exchanged_bytes > 0
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok", flush=True)


if __name__ == "__main__":
    main()
