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
    """CPU restatement of csrc/exchange_kernels.hip (same message layouts), built on the oracle's CSR expansion."""

    def pack_query(self, ids32, d_n, cap, q):
        n = ids32.numel()
        q[:n] = ids32
        q[cap] = n if d_n is None else min(int(d_n), n)

    def serve_rows(self, rowptr, col, req, n_peers, cap, lo, hi, reply, stride, e_slot, status):
        req = req.view(n_peers, cap + 1)
        reply = reply.view(n_peers, stride)
        for p in range(n_peers):
            m = int(req[p, cap].clamp(0, cap))
            ids = req[p, :cap].long()
            owned = (torch.arange(cap) < m) & (ids >= lo) & (ids < hi)
            loc = torch.where(owned, ids - lo, torch.zeros_like(ids))
            lens = torch.where(owned, rowptr[loc + 1] - rowptr[loc], torch.zeros_like(ids))
            offs = torch.cumsum(lens, 0) - lens
            reply[p, :cap] = lens.to(torch.int32)
            reply[p, cap:2 * cap] = offs.to(torch.int32)
            if int(lens.sum()) > e_slot:
                status |= 1
                continue
            nodes = loc[owned].numpy()
            edges = O.get_neighborhoods(nodes, rowptr.numpy(), col.numpy().astype(np.int64))   # utils.py:74-82
            reply[p, 2 * cap:2 * cap + edges.shape[1]] = torch.from_numpy(edges[1].astype(np.int32))

    def recv_rows(self, back, stride, nodes32, bounds32, n_peers, e_cap, d_m, status):
        cap = nodes32.numel()
        m = cap if d_m is None else int(d_m.clamp(0, cap))
        back2 = back.view(n_peers, stride)
        owner = torch.bucketize(nodes32[:m].long(), bounds32[1:-1].long(), right=True)
        i = torch.arange(m)
        lens = back2[owner, i].long()
        start = owner * stride + 2 * cap + back2[owner, cap + i].long()
        eoff = torch.zeros(cap + 1, dtype=torch.int32)
        eoff[1:m + 1] = torch.cumsum(lens, 0).to(torch.int32)
        e = int(lens.sum())
        src = torch.zeros(e_cap, dtype=torch.int32); dst = torch.zeros(e_cap, dtype=torch.int32)
        if e > e_cap:
            status |= 1
            e = 0
        rep = torch.repeat_interleave(i, lens)[:e]
        within = torch.arange(e) - eoff[:m].long()[rep]
        src[:e] = nodes32[rep]
        dst[:e] = back.view(-1)[start[rep] + within]
        return src, dst, torch.tensor([int(lens.sum())], dtype=torch.int32), eoff

    def serve_features(self, X, req, n_peers, cap, lo, hi, reply, n_slot, status):
        F = X.shape[1]
        req = req.view(n_peers, cap + 1)
        reply = reply.view(n_peers, n_slot, F)
        for p in range(n_peers):
            m = int(req[p, cap].clamp(0, cap))
            ids = req[p, :m].long()
            run = ids[(ids >= lo) & (ids < hi)]
            if run.numel() > n_slot:
                status |= 2
                run = run[:n_slot]
            reply[p, :run.numel()] = X[run - lo]

    def assemble_features(self, back, F, n_slot, ids32, bounds32, n_peers, d_n, ind_code, epoch, d_epoch, num_ind):
        n_rows = ids32.numel()
        n = n_rows if d_n is None else int(d_n.clamp(0, n_rows))
        ids = ids32[:n].long()
        cuts = torch.searchsorted(ids, bounds32.long())
        owner = torch.bucketize(ids, bounds32[1:-1].long(), right=True)
        r = (torch.arange(n) - cuts[owner]).clamp(max=n_slot - 1)
        out = torch.zeros(n_rows, F + num_ind)
        out[:n, :F] = back.view(n_peers, n_slot, F)[owner, r]
        if num_ind:
            ep = int(d_epoch) & 0xffffff if d_epoch is not None else epoch
            code = ind_code[ids]
            live = ((code >> 8) & 0xffffff) == ep
            for j in range(num_ind):
                out[:n, F + j] = (((code >> j) & 1) * live).float()
        return out


    # ---- rows found again among the rows already received (csrc/exchange_kernels.hip: halo_positions_k, exchange_note_rows_k)
    def halo_positions(self, ids32, bounds32, n_peers, n_slot, d_n, ind_code, pos, code_pos):
        n = ids32.numel() if d_n is None else int(d_n.clamp(0, ids32.numel()))
        ids = ids32[:n].long()
        cuts = torch.searchsorted(ids, bounds32.long())
        owner = torch.bucketize(ids, bounds32[1:-1].long(), right=True)
        r = (torch.arange(n) - cuts[owner]).clamp(max=n_slot - 1)
        pos.zero_()
        pos[:n] = (owner * n_slot + r).to(torch.int32)

    def note_rows(self, loc, ids32, d_n, pos, base, node_map, batch, d_n_batch, idx_a, idx_b):
        n = ids32.numel() if d_n is None else int(d_n.clamp(0, ids32.numel()))
        ids = ids32[:n].long()
        if node_map is not None:
            nb = batch.numel() if d_n_batch is None else int(d_n_batch)
            row = node_map[ids].long()
            ok = (row >= 0) & (row < nb)
            ok[ok.clone()] &= batch[row[ok]].long() == ids[ok]
            loc[ids[ok]] = (base + pos[row[ok]]).to(torch.int32)
        else:
            loc[ids] = (base + pos[idx_b[idx_a[:n].long()].long()]).to(torch.int32)

    def gather_noted(self, rows, loc, ids32, d_n):
        n = ids32.numel() if d_n is None else int(d_n.clamp(0, ids32.numel()))
        out = torch.zeros(ids32.numel(), rows.shape[1])
        out[:n] = rows[loc[ids32[:n].long()].long()]
        return out


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(123)                      # same full graph on every rank
    N, F = 5003, 7
    ei = rng.integers(0, N - 5, (2, 40000))               # (the last five nodes have no edge at all: isolated)
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
        e_cap = torch.tensor([e + 13]); dist.all_reduce(e_cap, op=dist.ReduceOp.MAX)      # equal on all ranks
        src, dst, d_e = g.expand(torch.from_numpy(nodes.astype(np.int32)), int(e_cap))
        assert int(d_e.item()) == e, (rank, int(d_e.item()), e)
        assert np.array_equal(src[:e].numpy().astype(np.int64), ref[0]), rank      # query order preserved
        assert np.array_equal(dst[:e].numpy().astype(np.int64), ref[1]), rank      # ascending column inside a row
    # capacity-padded query: only the first *d_m entries count, the padding holds arbitrary (valid) ids
    nodes = qrng.permutation(N)[:200].astype(np.int64)
    m_true = 120 + 7 * rank
    ref = O.get_neighborhoods(nodes[:m_true], indptr, indices)
    e = ref.shape[1]
    e_cap = torch.tensor([e + 5]); dist.all_reduce(e_cap, op=dist.ReduceOp.MAX)
    src, dst, d_e, eoff = g.expand(torch.from_numpy(nodes.astype(np.int32)), int(e_cap),
                                   d_m=torch.tensor([m_true], dtype=torch.int32), want_eoff=True)
    assert int(d_e.item()) == e and int(eoff[m_true]) == e
    lens = (eoff[1:m_true + 1] - eoff[:m_true]).numpy()
    assert np.array_equal(lens, indptr[nodes[:m_true] + 1] - indptr[nodes[:m_true]])
    assert np.array_equal(src[:e].numpy().astype(np.int64), ref[0]) and np.array_equal(dst[:e].numpy().astype(np.int64), ref[1])
    ids = np.sort(qrng.permutation(N)[:300]).astype(np.int32)
    out = g.features(torch.from_numpy(ids), d_n=torch.tensor([211], dtype=torch.int32))
    assert out.shape == (300, F) and torch.equal(out[:211], X[torch.from_numpy(ids[:211]).long()])
    for n_ids in (500, 1, 0):
        ids = np.sort(qrng.permutation(N)[:n_ids]).astype(np.int32)
        out = g.features(torch.from_numpy(ids))
        assert out.shape == (n_ids, F)
        assert torch.equal(out, X[torch.from_numpy(ids).long()]), rank             # halo rows are bit copies
    # a halo slot that is too small raises the status word instead of truncating silently
    g.slot_rows_fixed = 64
    ids = np.sort(qrng.permutation(N)[:900]).astype(np.int32)
    g.features(torch.from_numpy(ids), d_n=torch.tensor([900], dtype=torch.int32))
    flag = g.status.clone(); dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    assert int(flag) & 2
    g.status.zero_(); g.slot_rows_fixed = None
    # slot calibration from observed traffic
    g.calibrating = True
    g.features(torch.from_numpy(ids), d_n=torch.tensor([900], dtype=torch.int32))
    slot = g.calibrate(margin=1.0)
    runs = np.diff(np.searchsorted(ids, np.asarray(b)))
    assert slot >= runs.max() and slot <= runs.max() * world + 128
    out = g.features(torch.from_numpy(ids), d_n=torch.tensor([900], dtype=torch.int32))
    assert int(g.status) == 0 and torch.equal(out, X[torch.from_numpy(ids).long()])
    g.slot_rows_fixed = None
    # ---- the classifier's rows found among the hops' fetches instead of requested again (round 4: 9 -> 7 collectives per step).
    # Three "hops" with their own ascending batches, kept side by side; the noted nodes — some through a relabel table checked
    # against the batch (with an id that is NOT a batch row and an ISOLATED node, whose row is replicated), some through
    # (kept position -> candidate -> batch row) index chains — come back bit for bit, on every rank of a real partition.
    deg = np.diff(indptr)
    iso = np.flatnonzero(deg == 0)
    assert g.can_reuse_rows and iso.size > 0 and torch.equal(g.iso_ids.long(), torch.from_numpy(iso))
    hops, cap = 3, 640
    want_ids, want_rows = [], []
    for hop in range(hops):
        n_live = 600 - 50 * hop - rank
        batch = np.sort(qrng.permutation(N)[:cap]).astype(np.int32)[:cap]
        batch[:n_live] = np.sort(batch[:n_live])
        bt = torch.from_numpy(batch)
        d_nb = torch.tensor([n_live], dtype=torch.int32)
        halo = g.fetch_halo(bt, d_n=d_nb, cap=cap, keep=(hop, hops))
        pos, _, n_slot = g.halo_positions(bt, d_nb, cap, tag="t%d" % hop)
        assert halo["base"] == hop * world * n_slot and halo["n_slot"] == n_slot
        if hop == 0:      # through the relabel table: 40 batch rows, one id outside the batch, one isolated node
            node_map = torch.full((N,), 12345, dtype=torch.int32)          # stale everywhere ...
            node_map[bt[:n_live].long()] = torch.arange(n_live, dtype=torch.int32)      # ... but on the batch rows
            outside = int(np.setdiff1d(np.arange(N), np.concatenate([batch[:n_live], iso]))[7])
            picks = np.concatenate([batch[:n_live][::15][:40], [outside, iso[0]]]).astype(np.int32)
            g.note_rows(halo, torch.from_numpy(picks), None, pos, node_map=node_map, batch=bt, d_n_batch=d_nb)
            want_ids += list(batch[:n_live][::15][:40]) + [int(iso[0])]
        else:             # through the index chain of a draw: kept position -> candidate -> batch row
            nbl = torch.from_numpy(qrng.permutation(n_live)[:200].astype(np.int32))          # candidate -> batch row
            kept_pos = torch.from_numpy(np.sort(qrng.permutation(200)[:64]).astype(np.int32))
            kept_ids = bt[nbl[kept_pos.long()].long()].contiguous()
            g.note_rows(halo, kept_ids, torch.tensor([60], dtype=torch.int32), pos, idx_a=kept_pos, idx_b=nbl)
            want_ids += [int(v) for v in kept_ids[:60]]
    want = torch.tensor(sorted(set(want_ids)), dtype=torch.int32)
    got = g.rows_from_kept(halo, want, torch.tensor([want.numel()], dtype=torch.int32))
    assert torch.equal(got, X[want.long()]), rank
    assert int(g.status) == 0
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
