"""Multi-GPU path (SURVEY §8e — new design, the reference is single-device): the graph and the
feature matrix are 1-D node-partitioned over the P GPUs of one node; mini-batches are data-parallel
(each rank samples its own batch with its own model replica); what a rank's hop needs from rows it
does not own travels by RCCL all-to-all(v) over xGMI (every peer pair has its own link, so the
exchange uses all 7 links of a GPU at once), and the three small models' gradients are all-reduced
once per optimiser step.

Per hop (one rank's view)
  expand(previous_nodes):   ids -> owners (all-to-all), owners cut their CSR rows with the same
                            frontier kernels the single-GPU path uses, row lengths + column lists
                            come back (all-to-all), and are re-ordered into query order.
  features(batch_nodes):    ascending ids are already grouped by owner; owners gather their rows
                            (gather kernel) and send them back (all-to-all of halo feature rows).
Everything after that (compaction, sampler GCN, draw, slicing) is local and identical to the
single-GPU step, so sampled sets and activations do not depend on P (halo rows are bit copies).

The exchange layer is written against `torch.distributed` only (backend "nccl" == RCCL on ROCm;
"gloo" in the CPU tests) and against a tiny `local_ops` interface, so that tests can drive it on the
CPU with an oracle-backed double while production uses the HIP kernels.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _lib


class HipLocalOps:
    """Production implementation of the local pieces: the gfx950 kernels."""

    def offsets(self, rowptr, nodes32):
        from . import ops
        return ops.frontier_offsets(rowptr, nodes32)

    def expand(self, rowptr, col, nodes32, eoff, e_cap, want_pos=False):
        from . import ops
        return ops.frontier_expand(rowptr, col, nodes32, eoff, e_cap, want_pos=want_pos)

    def gather_rows(self, X, ids32):
        from . import ops
        return ops.gather_rows(X, ids32)

    def take(self, table32, keys32):
        from . import ops
        return ops.tensormap_map(table32, keys32)


def partition_bounds(num_nodes: int, world: int) -> List[int]:
    """Contiguous, near-equal node ranges: rank p owns [bounds[p], bounds[p+1])."""
    return [(num_nodes * p) // world for p in range(world + 1)]


class GraphScratch:
    """Per-rank scratch tables over the GLOBAL id space (see graph.DeviceGraph)."""

    def _alloc_scratch(self, num_nodes: int, device):
        W = (num_nodes + 63) // 64
        W1 = (W + 63) // 64
        self.num_nodes = int(num_nodes)
        self.device = torch.device(device)
        self.bits = torch.zeros(W, dtype=torch.int64, device=device)
        self.bits1 = torch.zeros(W1, dtype=torch.int64, device=device)
        self.prev_bits = torch.zeros(W, dtype=torch.int64, device=device)
        self.node_map = torch.empty(num_nodes, dtype=torch.int32, device=device)
        self.mult = torch.zeros(num_nodes, dtype=torch.int32, device=device)
        self.ind_code = torch.zeros(num_nodes, dtype=torch.int32, device=device)
        self.status = torch.zeros(1, dtype=torch.int32, device=device)

    def check_status(self, what: str = "hop pipeline"):
        s = int(self.status.item())
        if s:
            self.status.zero_()
            bits = [n for b, n in ((1, "edge buffer overflow"), (2, "node buffer overflow"), (4, "index out of range"))
                    if s & b]
            raise _lib.GrapesHipError(f"{what}: " + ", ".join(bits))


class PartitionedGraph(GraphScratch):
    """One rank's shard of a 1-D node-partitioned graph + feature matrix.

    rowptr_local int64[n_loc+1] (rebased to 0), col_local int32[nnz_loc] (GLOBAL column ids),
    X_local fp32[n_loc, F]; bounds = partition_bounds(N, P)."""

    def __init__(self, rowptr_local: torch.Tensor, col_local: torch.Tensor, X_local: torch.Tensor,
                 bounds: Sequence[int], rank: int, world: int, group=None, local_ops=None, max_degree: int = 0,
                 alloc_scratch: bool = True):
        self.rowptr, self.col, self.X = rowptr_local.contiguous(), col_local.contiguous(), X_local.contiguous()
        self.bounds = [int(b) for b in bounds]
        self.rank, self.world, self.group = rank, world, group
        self.lo, self.hi = self.bounds[rank], self.bounds[rank + 1]
        assert self.rowptr.numel() == self.hi - self.lo + 1
        dev = self.rowptr.device
        self.ops = local_ops or HipLocalOps()
        self.inner = torch.tensor(self.bounds[1:-1], dtype=torch.int64, device=dev)
        self.bounds_t = torch.tensor(self.bounds, dtype=torch.int64, device=dev)
        self.max_degree = max_degree
        self.feature_dim = self.X.shape[1]
        self.exchanged_bytes = 0            # payload this rank sent, for reporting
        if alloc_scratch:
            self._alloc_scratch(self.bounds[-1], dev)
        else:
            self.num_nodes, self.device = self.bounds[-1], dev

    # ------------------------------------------------------------------ collectives
    def _counts(self, send_counts: torch.Tensor):
        """all-to-all of per-peer counts; returns (send list, recv list) on the host (one sync)."""
        recv = torch.empty_like(send_counts)
        dist.all_to_all_single(recv, send_counts, group=self.group)
        both = torch.stack([send_counts, recv]).tolist()
        return both[0], both[1]

    def _a2a(self, send: torch.Tensor, in_splits: List[int], out_splits: List[int], width: int = 1) -> torch.Tensor:
        out = torch.empty((sum(out_splits) * width,), dtype=send.dtype, device=send.device)
        dist.all_to_all_single(out, send.reshape(-1), output_split_sizes=[c * width for c in out_splits],
                               input_split_sizes=[c * width for c in in_splits], group=self.group)
        self.exchanged_bytes += send.numel() * send.element_size()
        return out

    # ------------------------------------------------------------------ A1 across shards
    def expand(self, nodes32: torch.Tensor, e_cap: int, d_m: Optional[torch.Tensor] = None):
        """Same contract as frontier_offsets + frontier_expand on the full graph: (src, dst, d_e) with
        src/dst of capacity e_cap, edges in query order then ascending column.  Only the first *d_m entries
        of nodes32 are queried when d_m (device int32[1]) is given.

        Three all-to-alls and ONE host read: requests travel in fixed-size slots (every rank passes the same
        nodes32.numel(), padding = -1), row lengths come back in the same slots, and only the concatenated
        column lists need sizes on the host.  All ranks must call with equally sized `nodes32`."""
        dev, P = nodes32.device, self.world
        cap = nodes32.numel()
        if cap == 0:
            return (torch.empty(e_cap, dtype=torch.int32, device=dev), torch.empty(e_cap, dtype=torch.int32, device=dev),
                    torch.zeros(1, dtype=torch.int32, device=dev))
        nodes = nodes32.long()
        ar = torch.arange(cap, device=dev)
        owner = torch.bucketize(nodes, self.inner, right=True)
        if d_m is not None:
            owner = torch.where(ar < d_m.long(), owner, torch.full_like(owner, P))           # bucket P = not queried
        order = torch.argsort(owner, stable=True)
        s_owner = owner[order]
        counts = torch.bincount(owner, minlength=P + 1)
        seg_start = torch.cumsum(counts, 0) - counts
        rank_in = ar - seg_start[s_owner]
        trash = P * cap
        slot = torch.where(s_owner < P, s_owner * cap + rank_in, torch.full_like(s_owner, trash))
        send = torch.full((P * cap + 1,), -1, dtype=torch.int64, device=dev)
        send[slot] = nodes[order]
        send[trash] = -1
        req = torch.empty(P * cap, dtype=torch.int64, device=dev)
        dist.all_to_all_single(req, send[:P * cap].contiguous(), group=self.group)           # (1) ids, fixed slots
        valid = req >= 0
        local = torch.where(valid, req - self.lo, torch.zeros_like(req)).to(torch.int32)
        eoff_raw, _ = self.ops.offsets(self.rowptr, local)
        lens = ((eoff_raw[1:] - eoff_raw[:-1]) * valid.to(torch.int32)).contiguous()        # 0 for padding slots
        eoff = torch.zeros(P * cap + 1, dtype=torch.int32, device=dev)
        torch.cumsum(lens, 0, out=eoff[1:])
        tot_peer = (eoff[cap::cap] - eoff[0:P * cap:cap]).long()                            # edges served per peer
        lens_back = torch.empty(P * cap, dtype=torch.int32, device=dev)
        dist.all_to_all_single(lens_back, lens, group=self.group)                            # (2) lengths, same slots
        recv_tot = lens_back.view(P, cap).sum(dim=1)
        sizes = torch.cat([tot_peer, recv_tot]).tolist()                                     # the one host read
        et_s, et_r = sizes[:P], sizes[P:]
        e_serv, e_tot = sum(et_s), sum(et_r)
        if e_serv > 0:
            _, dst_serv, _ = self.ops.expand(self.rowptr, self.col, local, eoff, e_serv)
        else:
            dst_serv = torch.empty(0, dtype=torch.int32, device=dev)
        dst_back = self._a2a(dst_serv, et_s, et_r)                                           # (3) column lists
        if e_tot > e_cap:
            raise _lib.GrapesHipError(f"frontier of {e_tot} edges exceeds e_cap={e_cap}")
        # received data is a CSR over the slots (+ one empty trash row); read it back in query order
        rowptr_recv = torch.zeros(P * cap + 2, dtype=torch.int64, device=dev)
        torch.cumsum(lens_back.long(), 0, out=rowptr_recv[1:P * cap + 1])
        rowptr_recv[P * cap + 1] = rowptr_recv[P * cap]
        inv = torch.empty(cap, dtype=torch.int64, device=dev)
        inv[order] = slot
        inv32 = inv.to(torch.int32)
        eoff2, d_e = self.ops.offsets(rowptr_recv, inv32)
        src_full = torch.empty(e_cap, dtype=torch.int32, device=dev)
        if e_tot > 0:
            _, dst, pos = self.ops.expand(rowptr_recv, dst_back, inv32, eoff2, e_cap, want_pos=True)
            src_full[:e_tot] = self.ops.take(nodes32, pos[:e_tot].contiguous())
        else:
            dst = torch.empty(e_cap, dtype=torch.int32, device=dev)
        self.exchanged_bytes += 12 * P * cap
        return src_full, dst, d_e

    # ------------------------------------------------------------------ halo feature rows
    def features(self, ids32_sorted: torch.Tensor, d_n: Optional[torch.Tensor] = None) -> torch.Tensor:
        """X[ids] for ASCENDING global ids (batch_nodes / all_nodes are): fp32[n, F], n = *d_n if given."""
        F = self.feature_dim
        ids = ids32_sorted.long()
        if d_n is not None:   # padding -> sentinel beyond every partition bound (keeps the list ascending)
            ids = torch.where(torch.arange(ids.numel(), device=ids.device) < d_n.long(), ids,
                              torch.full_like(ids, self.bounds[-1]))
        cuts = torch.searchsorted(ids, self.bounds_t)
        sc, rc = self._counts((cuts[1:] - cuts[:-1]))
        req = self._a2a(ids[:sum(sc)].contiguous(), sc, rc)
        rows = self.ops.gather_rows(self.X, (req - self.lo).to(torch.int32))
        back = self._a2a(rows, rc, sc, width=F)
        return back.view(-1, F)


def shard_full_graph(rowptr: torch.Tensor, col: torch.Tensor, X: torch.Tensor, rank: int, world: int, group=None,
                     local_ops=None, max_degree: int = 0) -> PartitionedGraph:
    """Cuts this rank's shard out of a full (replicated) CSR + feature matrix."""
    N = rowptr.numel() - 1
    b = partition_bounds(N, world)
    lo, hi = b[rank], b[rank + 1]
    rp = (rowptr[lo:hi + 1] - rowptr[lo]).clone()
    cl = col[int(rowptr[lo]):int(rowptr[hi])].clone()
    return PartitionedGraph(rp, cl, X[lo:hi].clone(), b, rank, world, group, local_ops, max_degree)


def make_grad_sync(world: int, group=None):
    """All-reduce (mean) of the gradients of a parameter list, as one flat bucket (RCCL; the three
    GRAPES models together are < 2 MB, i.e. latency-bound: one collective per optimiser step)."""
    def sync(params):
        gs = [p.grad for p in params if p.grad is not None]
        if not gs:
            return
        flat = torch.cat([g.reshape(-1) for g in gs])
        dist.all_reduce(flat, group=group)
        flat /= world
        o = 0
        for g in gs:
            n = g.numel()
            g.copy_(flat[o:o + n].view_as(g))
            o += n
    return sync
