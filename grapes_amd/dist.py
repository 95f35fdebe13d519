"""Multi-GPU path (SURVEY §8e — new design, the reference is single-device): the graph and the
feature matrix are 1-D node-partitioned over the P GPUs of one node; mini-batches are data-parallel
(each rank samples its own batch with its own model replica); what a rank's hop needs from rows it
does not own travels over xGMI by RCCL (every peer pair has its own link, so an all-to-all uses all
7 links of a GPU at once), and the three small models' gradients are all-reduced once per
optimiser step.

Per hop (one rank's view)
  expand(previous_nodes):   all-gather of the query lists (<= B+K ids per rank), the owner of an id cuts
                            its CSR row into the requester's reply slot, ONE all-to-all brings the slots
                            back, the requester reads them in query order.
  features(batch_nodes):    all-gather of the ascending id lists; an owner's ids are one contiguous run of
                            each list, which it gathers into that peer's slot; ONE all-to-all of halo
                            feature rows.
Every message has a fixed capacity that is equal on all ranks and carries its live count inside, so a
hop never reads a size on the host: the step between two collectives is captured as a hipGraph
segment (capture.SegmentedGraph) and the collectives are launched between the segments.
Everything after the exchange (compaction, sampler GCN, draw, slicing) is local and identical to the
single-GPU step, so sampled sets and activations do not depend on P (halo rows are bit copies).

The exchange layer is written against `torch.distributed` only (backend "nccl" == RCCL on ROCm;
"gloo" in the CPU tests) and against a small `local_ops` interface, so that tests can drive it on the
CPU with an oracle-backed double while production uses the HIP kernels (csrc/exchange_kernels.hip).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _lib
from .graph import EpochSpace


class HipLocalOps:
    """Production implementation of the pieces either side of the collectives: the gfx950 kernels."""

    def pack_query(self, ids32, d_n, cap, q):
        from . import ops
        ops.exchange_pack_query(ids32, d_n, cap, q)

    def serve_rows(self, rowptr, col, req, n_peers, cap, lo, hi, reply, stride, e_slot, status):
        from . import ops
        ops.exchange_serve_rows(rowptr, col, req, n_peers, cap, lo, hi, reply, stride, e_slot, status=status)

    def recv_rows(self, back, stride, nodes32, bounds32, n_peers, e_cap, d_m, status):
        from . import ops
        return ops.exchange_recv_rows(back, stride, nodes32, bounds32, n_peers, e_cap, d_m=d_m, status=status)

    def serve_features(self, X, req, n_peers, cap, lo, hi, reply, n_slot, status):
        from . import ops
        ops.exchange_serve_features(X, req, n_peers, cap, lo, hi, reply, n_slot, status=status)

    def assemble_features(self, back, F, n_slot, ids32, bounds32, n_peers, d_n, ind_code, epoch, d_epoch, num_ind):
        from . import ops
        return ops.exchange_assemble_features(back, F, n_slot, ids32, bounds32, n_peers, d_n=d_n, ind_code=ind_code,
                                              epoch=epoch, d_epoch=d_epoch, num_ind=num_ind)

    in_place_halo = True      # the fused gather-SpMM can read the exchanged rows where they arrive (halo_positions)

    def note_rows(self, loc, ids32, d_n, pos, base, node_map, batch, d_n_batch, idx_a, idx_b):
        from . import ops
        ops.exchange_note_rows(loc, ids32, pos, base, d_n=d_n, node_map=node_map, batch=batch, d_n_batch=d_n_batch, idx_a=idx_a,
                               idx_b=idx_b)

    def gather_noted(self, rows, loc, ids32, d_n):
        from . import ops
        where = ops.tensormap_map(loc, ids32, d_n=d_n)
        return ops.gather_rows(rows, where, d_n=d_n)

    def halo_positions(self, ids32, bounds32, n_peers, n_slot, d_n, ind_code, pos, code_pos):
        from . import ops
        return ops.exchange_halo_positions(ids32, bounds32, n_peers, n_slot, d_n=d_n, ind_code=ind_code, pos=pos, code_pos=code_pos)


def partition_bounds(num_nodes: int, world: int) -> List[int]:
    """Contiguous, near-equal node ranges: rank p owns [bounds[p], bounds[p+1])."""
    return [(num_nodes * p) // world for p in range(world + 1)]


class GraphScratch(EpochSpace):
    """Per-rank scratch tables over the GLOBAL id space (see graph.DeviceGraph)."""

    def _alloc_scratch(self, num_nodes: int, device):
        W = (num_nodes + 63) // 64
        W1 = (W + 63) // 64
        self.num_nodes = int(num_nodes)
        self.device = torch.device(device)
        self.bits = torch.zeros(W, dtype=torch.int64, device=device)
        self.bits1 = None
        self.prev_bits = torch.zeros(W, dtype=torch.int64, device=device)
        self.prev_bits_b = torch.zeros(W, dtype=torch.int64, device=device)   # (the captured step alternates the two per hop)
        self.node_map = torch.empty(num_nodes, dtype=torch.int32, device=device)
        self.mult = torch.zeros(num_nodes, dtype=torch.int32, device=device)
        self.ind_code = torch.zeros(num_nodes, dtype=torch.int32, device=device)
        self.status = torch.zeros(1, dtype=torch.int32, device=device)

    def check_status(self, what: str = "hop pipeline"):
        s = int(self.status.item())
        if s:
            self.status.zero_()
            bits = [n for b, n in ((1, "edge buffer overflow"), (2, "node buffer overflow"), (4, "index out of range"),
                                      (8, "one-launch scan timed out"))
                    if s & b]
            raise _lib.GrapesHipError(f"{what}: " + ", ".join(bits))


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


class PartitionedGraph(GraphScratch):
    """One rank's shard of a 1-D node-partitioned graph + feature matrix.

    rowptr_local int64[n_loc+1] (rebased to 0), col_local int32[nnz_loc] (GLOBAL column ids),
    X_local fp32[n_loc, F]; bounds = partition_bounds(N, P).

    Slot sizes: a reply slot of the row exchange holds `e_cap` columns (a requester never receives more than
    e_cap edges in total, so the slot cannot overflow before the requester does); a reply slot of the feature
    exchange holds `halo_slot_rows(cap)` rows = cap·slot_factor/P (the ids of a frontier spread over the owners),
    or what `calibrate()` measured during warm-up.  A slot that is too small raises the device status word."""

    def __init__(self, rowptr_local: torch.Tensor, col_local: torch.Tensor, X_local: torch.Tensor,
                 bounds: Sequence[int], rank: int, world: int, group=None, local_ops=None, max_degree: int = 0,
                 alloc_scratch: bool = True, slot_factor: float = 2.0, full_rowptr: Optional[torch.Tensor] = None,
                 full_col: Optional[torch.Tensor] = None,
                 isolated: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        self.rowptr, self.col, self.X = rowptr_local.contiguous(), col_local.contiguous(), X_local.contiguous()
        # Replicated adjacency (the default of shard_full_graph): every rank keeps the WHOLE CSR (ogbn-products 0.5 GB,
        # papers100M symmetrised 13.8 GB of a 288 GB HBM) and only the feature matrix — the bulk: 57 GB for papers100M — is
        # 1-D partitioned.  get_neighborhoods is then local (the single-GPU kernels) and the only exchange of a hop is
        # the halo feature fetch: 2 collectives per hop instead of 4.
        self.rowptr_full = None if full_rowptr is None else full_rowptr.contiguous()
        self.col_full = None if full_col is None else full_col.contiguous()
        self.adjacency_replicated = self.rowptr_full is not None
        # isolated = (ids int32 ascending, rows fp32[len(ids), F]): the nodes WITHOUT any edge and their feature rows, replicated on
        # every rank.  They are never batch rows of a hop (get_neighborhoods returns nothing for them), so a hop's exchange never
        # brings their features — yet an isolated TARGET is one of all_nodes (main.py:164,252).  With them at hand the
        # classifier's features are all among rows this rank already holds (rows_from_kept); without, they are requested.
        self.iso_ids = None if isolated is None else isolated[0].to(torch.int32).contiguous()
        self.iso_rows = None if isolated is None else isolated[1].contiguous()
        self.bounds = [int(b) for b in bounds]
        self.rank, self.world, self.group = rank, world, group
        self.lo, self.hi = self.bounds[rank], self.bounds[rank + 1]
        assert self.rowptr.numel() == self.hi - self.lo + 1
        assert self.bounds[-1] < 2 ** 31
        dev = self.rowptr.device
        self.ops = local_ops or HipLocalOps()
        self.bounds32 = torch.tensor(self.bounds, dtype=torch.int32, device=dev)
        self.max_degree = max_degree
        self.feature_dim = self.X.shape[1]
        self.exchanged_bytes = 0            # payload capacity this rank sent, for reporting
        self.slot_factor = float(slot_factor)
        self.slot_rows_fixed: Optional[int] = None
        self.calibrating = False
        self.peak_rows = torch.zeros(1, dtype=torch.int64, device=dev)     # largest per-peer run seen (calibration)
        self.run_collective: Callable[[Callable[[], None]], None] = lambda fn: fn()   # capture.SegmentedGraph hooks in
        self._bufs: Dict[Tuple, torch.Tensor] = {}
        if alloc_scratch:
            self._alloc_scratch(self.bounds[-1], dev)
        else:
            self.num_nodes, self.device = self.bounds[-1], dev
            self.status = torch.zeros(1, dtype=torch.int32, device=dev)

    # ------------------------------------------------------------------ buffers / collectives
    def _buf(self, tag: str, numel: int, dtype) -> torch.Tensor:
        """Persistent message buffers (one per tag and size): their addresses are baked into captured segments and
        into the collectives launched between them.  A buffer is consumed before the next call of the same tag."""
        key = (tag, int(numel), dtype)
        b = self._bufs.get(key)
        if b is None:
            b = torch.zeros(int(numel), dtype=dtype, device=self.rowptr.device)
            self._bufs[key] = b
        return b

    def _all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        self.run_collective(lambda: dist.all_gather_into_tensor(out, inp, group=self.group))
        self.exchanged_bytes += inp.numel() * inp.element_size() * max(self.world - 1, 1)

    def _all_to_all(self, out: torch.Tensor, inp: torch.Tensor):
        self.run_collective(lambda: dist.all_to_all_single(out, inp, group=self.group))
        self.exchanged_bytes += inp.numel() * inp.element_size()

    def _common_cap(self, n: int) -> int:
        """Eager callers with exact-size lists: the capacity all ranks agree on (one tiny all-reduce + host read)."""
        if self.world == 1:
            return n
        t = torch.tensor([n], dtype=torch.int64, device=self.rowptr.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def _query(self, tag: str, ids32: torch.Tensor, d_n: Optional[torch.Tensor], cap: int) -> torch.Tensor:
        """[cap ids | live count] in a persistent buffer (ids beyond the list are padding)."""
        q = self._buf(tag, cap + 1, torch.int32)
        self.ops.pack_query(ids32, d_n, cap, q)            # one kernel (never copy_(): that becomes a D2D memcpy node)
        return q

    # ------------------------------------------------------------------ A1 across shards
    def expand(self, nodes32: torch.Tensor, e_cap: int, d_m: Optional[torch.Tensor] = None, cap: Optional[int] = None,
               want_eoff: bool = False):
        """Same contract as frontier_offsets + frontier_expand on the full graph: (src, dst, d_e) with
        src/dst of capacity e_cap, edges in query order then ascending column.  Only the first *d_m entries
        of nodes32 are queried when d_m (device int32[1]) is given.  One all-gather + one all-to-all, no host read.
        All ranks must use the same e_cap and the same capacity: `cap`, else nodes32.numel() when d_m is given
        (the captured step), else an agreed maximum (eager callers with exact-size lists)."""
        dev, P = nodes32.device, self.world
        if self.adjacency_replicated:                      # local: the single-GPU kernels on this rank's copy of the CSR
            from . import ops
            eoff, d_e = ops.frontier_offsets(self.rowptr_full, nodes32, d_m=d_m)
            src, dst, _ = ops.frontier_expand(self.rowptr_full, self.col_full, nodes32, eoff, e_cap, d_m=d_m, status=self.status)
            return (src, dst, d_e, eoff) if want_eoff else (src, dst, d_e)
        if cap is None:
            cap = nodes32.numel() if d_m is not None else self._common_cap(nodes32.numel())
        if cap == 0:
            z = torch.zeros(1, dtype=torch.int32, device=dev)
            out = (torch.empty(e_cap, dtype=torch.int32, device=dev), torch.empty(e_cap, dtype=torch.int32, device=dev), z)
            return out + (z.clone(),) if want_eoff else out
        q = self._query("row_q", nodes32, d_m, cap)
        req = self._buf("row_req", P * (cap + 1), torch.int32)
        self._all_gather(req, q)                                                          # (1) query lists
        stride = 2 * cap + e_cap
        reply = self._buf("row_reply", P * stride, torch.int32)
        self.ops.serve_rows(self.rowptr, self.col, req, P, cap, self.lo, self.hi, reply, stride, e_cap, self.status)
        back = self._buf("row_back", P * stride, torch.int32)
        self._all_to_all(back, reply)                                                     # (2) [len | off | columns]
        src, dst, d_e, eoff = self.ops.recv_rows(back, stride, q[:cap], self.bounds32, P, e_cap, q[cap:], self.status)
        return (src, dst, d_e, eoff) if want_eoff else (src, dst, d_e)

    # ------------------------------------------------------------------ halo feature rows
    def halo_slot_rows(self, cap: int) -> int:
        if self.world == 1:
            return cap
        if self.slot_rows_fixed is not None:
            return min(cap, self.slot_rows_fixed)
        return min(cap, _round_up(int(cap * self.slot_factor / self.world) + 1, 64))

    def fetch_halo(self, ids32_sorted: torch.Tensor, d_n: Optional[torch.Tensor] = None, cap: Optional[int] = None,
                   keep: Optional[Tuple[int, int]] = None):
        """Brings the feature rows of ASCENDING global ids to this rank.  Returns an opaque handle for
        `assemble()`; the rows stay valid until the next fetch_halo of the same capacity — or, with keep = (slot, slots), until
        the next fetch into the same slot of ONE buffer that holds `slots` fetches of this capacity side by side
        (handle["base"] = the slot's first row in handle["all"], viewed as [slots * P * n_slot, F]: rows_from_kept)."""
        P, F = self.world, self.feature_dim
        if cap is None:
            cap = ids32_sorted.numel() if d_n is not None else self._common_cap(ids32_sorted.numel())
        n_slot = self.halo_slot_rows(cap)
        q = self._query("feat_q", ids32_sorted, d_n, cap)
        if self.calibrating:
            live = torch.arange(cap, device=q.device) < q[cap:].long()
            ids = torch.where(live, q[:cap].long(), torch.full((cap,), self.bounds[-1], dtype=torch.int64, device=q.device))
            cuts = torch.searchsorted(ids, self.bounds32.long())
            self.peak_rows = torch.maximum(self.peak_rows, (cuts[1:] - cuts[:-1]).max().reshape(1))
        req = self._buf("feat_req", P * (cap + 1), torch.int32)
        self._all_gather(req, q)                                                          # (1) ascending id lists
        reply = self._buf("feat_reply", P * n_slot * F, torch.float32)
        self.ops.serve_features(self.X, req, P, cap, self.lo, self.hi, reply, n_slot, self.status)
        if keep is None:
            back = self._buf("feat_back", P * n_slot * F, torch.float32)
            extra = {}
        else:
            slot, slots = keep
            allb = self._kept_buffer(slots, n_slot)
            back = allb[slot * P * n_slot * F:(slot + 1) * P * n_slot * F]
            extra = dict(all=allb, base=slot * P * n_slot)
        self._all_to_all(back, reply)                                                     # (2) halo feature rows
        return dict(back=back, n_slot=n_slot, ids=q[:cap], d_n=q[cap:], cap=cap, **extra)

    def halo_positions(self, ids32_sorted: torch.Tensor, d_n: Optional[torch.Tensor], cap: int,
                       ind_code: Optional[torch.Tensor] = None, tag: str = "hop"):
        """Where the rows of ASCENDING `ids32_sorted` will sit in fetch_halo(ids, cap=cap)["back"] viewed as [P * n_slot, F], and
        their indicator words at those positions: (pos int32[len(ids)], code_pos | None, n_slot).  Depends on the id list only —
        the hop's graph build can write its head records with head_ids = pos before the exchange runs."""
        n_slot = self.halo_slot_rows(cap)
        pos = self._buf("halo_pos_" + tag, ids32_sorted.numel(), torch.int32)
        code_pos = self._buf("halo_code_" + tag, self.world * n_slot, torch.int32) if ind_code is not None else None
        self.ops.halo_positions(ids32_sorted, self.bounds32, self.world, n_slot, d_n, ind_code, pos, code_pos)
        return pos, code_pos, n_slot

    def assemble(self, halo, n_rows: Optional[int] = None, ind_code: Optional[torch.Tensor] = None, epoch: int = 0,
                 d_epoch: Optional[torch.Tensor] = None, num_ind: int = 0) -> torch.Tensor:
        """[X[ids] | indicators(ids)] fp32[n_rows or cap, F + num_ind] from a fetch_halo handle."""
        ids = halo["ids"] if n_rows is None else halo["ids"][:n_rows]
        return self.ops.assemble_features(halo["back"], self.feature_dim, halo["n_slot"], ids, self.bounds32, self.world,
                                          halo["d_n"], ind_code if num_ind else None, epoch, d_epoch, num_ind)

    # ---- rows this rank has already received in the step, found again instead of requested again
    @property
    def can_reuse_rows(self) -> bool:
        return self.iso_ids is not None

    def _kept_buffer(self, slots: int, n_slot: int) -> torch.Tensor:
        """[slots fetches of P * n_slot rows | the isolated nodes' rows] x F, one allocation per (slots, n_slot); the isolated
        rows and their (static) locations are written when it is created — outside any capture (the warm-up steps)."""
        P, F = self.world, self.feature_dim
        n_iso = 0 if self.iso_ids is None else int(self.iso_ids.numel())
        key = ("feat_back_kept", (slots * P * n_slot + n_iso) * F, torch.float32)
        b = self._bufs.get(key)
        if b is None:
            b = torch.zeros(key[1], dtype=torch.float32, device=self.rowptr.device)
            self._bufs[key] = b
            if n_iso:
                b[slots * P * n_slot * F:].copy_(self.iso_rows.reshape(-1))
                self.row_locations()[self.iso_ids.long()] = (slots * P * n_slot +
                                                             torch.arange(n_iso, dtype=torch.int32, device=b.device))
        return b

    def row_locations(self) -> torch.Tensor:
        """int32[N]: for the nodes noted with note_rows since their fetch, the row of the kept buffer that holds their features"""
        t = getattr(self, "_loc", None)
        if t is None:
            t = torch.zeros(self.bounds[-1], dtype=torch.int32, device=self.rowptr.device)
            self._loc = t
        return t

    def note_rows(self, halo, ids32: torch.Tensor, d_n: Optional[torch.Tensor], pos: torch.Tensor, node_map=None, batch=None,
                  d_n_batch=None, idx_a=None, idx_b=None):
        """The feature rows of ids[0 .. *d_n) are batch rows of the kept fetch `halo` (fetch_halo(keep=...)): row(i) =
        node_map[ids[i]] (checked against `batch`: an id that is no batch row — a target without edges — keeps its static
        location) or idx_b[idx_a[i]] of its batch, pos = halo_positions of that batch."""
        self.ops.note_rows(self.row_locations(), ids32, d_n, pos, halo["base"], node_map, batch, d_n_batch, idx_a, idx_b)

    def rows_from_kept(self, halo, ids32: torch.Tensor, d_n: Optional[torch.Tensor]) -> torch.Tensor:
        """X[ids] fp32[len(ids), F] for nodes whose rows were noted: a local gather, no exchange."""
        return self.ops.gather_noted(halo["all"].view(-1, self.feature_dim), self.row_locations(), ids32, d_n)

    def features(self, ids32_sorted: torch.Tensor, d_n: Optional[torch.Tensor] = None) -> torch.Tensor:
        """X[ids] for ASCENDING global ids (batch_nodes / all_nodes are): fp32[len(ids), F]."""
        cap = ids32_sorted.numel() if d_n is not None else self._common_cap(ids32_sorted.numel())
        if cap == 0:
            return torch.empty((0, self.feature_dim), dtype=torch.float32, device=ids32_sorted.device)
        return self.assemble(self.fetch_halo(ids32_sorted, d_n, cap=cap), n_rows=ids32_sorted.numel())

    # ------------------------------------------------------------------ slot calibration
    def calibrate(self, margin: float = 2.0):
        """After warm-up steps run with `calibrating = True`: fixes the halo slot size at margin x the largest
        per-peer run any rank saw (one host read + one all-reduce, outside the timed / captured region)."""
        peak = self.peak_rows.clone()
        if self.world > 1:
            dist.all_reduce(peak, op=dist.ReduceOp.MAX, group=self.group)
        self.slot_rows_fixed = _round_up(int(int(peak.item()) * margin) + 64, 64)
        self.calibrating = False
        return self.slot_rows_fixed


def shard_full_graph(rowptr: torch.Tensor, col: torch.Tensor, X: torch.Tensor, rank: int, world: int, group=None,
                     local_ops=None, max_degree: int = 0, slot_factor: float = 2.0,
                     replicate_adjacency: bool = False) -> PartitionedGraph:
    """Cuts this rank's shard out of a full CSR + feature matrix.  replicate_adjacency: keep the whole CSR on every rank
    (features only are partitioned; see PartitionedGraph)."""
    N = rowptr.numel() - 1
    b = partition_bounds(N, world)
    lo, hi = b[rank], b[rank + 1]
    rp = (rowptr[lo:hi + 1] - rowptr[lo]).clone()
    cl = col[int(rowptr[lo]):int(rowptr[hi])].clone()
    iso = (rowptr[1:] == rowptr[:-1]).nonzero().view(-1)          # nodes without an edge: their rows are replicated (see the class)
    return PartitionedGraph(rp, cl, X[lo:hi].clone(), b, rank, world, group, local_ops, max_degree,
                            slot_factor=slot_factor, full_rowptr=rowptr if replicate_adjacency else None,
                            full_col=col if replicate_adjacency else None,
                            isolated=(iso.to(torch.int32), X[iso].clone()))


class GradSync:
    """All-reduce (mean) of the gradients of a parameter list as ONE flat bucket (RCCL; the three GRAPES models
    together are < 2 MB, i.e. latency-bound: one collective per optimiser step).  The bucket is persistent and the
    collective goes through `run_collective`, so the packing / unpacking kernels can live in captured segments."""

    def __init__(self, world: int, group=None):
        self.world, self.group = world, group
        self.run_collective: Callable[[Callable[[], None]], None] = lambda fn: fn()
        self._flat: Dict[Tuple, torch.Tensor] = {}
        self._bucket = None        # (parameter ids, flat tensor, element offsets) once make_bucket() has placed the gradients

    def make_bucket(self, params):
        """Allocates the gradients of `params` (none may exist yet) as 16-byte aligned views of ONE flat tensor, in this order:
        the all-reduce then runs on the gradients where the backward kernels wrote them — no packing / unpacking launches
        around the collective (two multi-tensor copies per step otherwise)."""
        params = list(params)
        if not params or any(p.grad is not None for p in params):
            return False
        offs, o = [], 0
        for p in params:
            offs.append(o)
            o += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(o, dtype=params[0].dtype, device=params[0].device)
        for p, off in zip(params, offs):
            p.grad = flat[off:off + p.numel()].view_as(p)
        self._bucket = (tuple(id(p) for p in params), flat, offs)
        return True

    def _all_reduce(self, flat: torch.Tensor):
        dist.all_reduce(flat, group=self.group)

    def __call__(self, params):
        params = [p for p in params if p.grad is not None]
        gs = [p.grad for p in params]
        if not gs:
            return
        key = tuple(id(p) for p in params)
        if self._bucket is not None and self._bucket[0] == key:
            _, flat, offs = self._bucket
            es = flat.element_size()
            if all(g.data_ptr() == flat.data_ptr() + off * es and g.is_contiguous() for g, off in zip(gs, offs)):
                self.run_collective(lambda: self._all_reduce(flat))      # the gradients ARE the bucket
                if self.world > 1:
                    flat.div_(self.world)
                return
        flat = self._flat.get(key)
        if flat is None:
            flat = torch.empty(sum(g.numel() for g in gs), dtype=gs[0].dtype, device=gs[0].device)
            self._flat[key] = flat
        views, o = [], 0
        for g in gs:
            views.append(flat[o:o + g.numel()].view_as(g))
            o += g.numel()
        torch._foreach_copy_(views, gs)
        self.run_collective(lambda: self._all_reduce(flat))
        flat.div_(self.world)
        torch._foreach_copy_(gs, views)


def make_grad_sync(world: int, group=None) -> GradSync:
    return GradSync(world, group)
