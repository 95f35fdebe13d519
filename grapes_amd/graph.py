"""Device-resident graph: the HBM counterpart of the SciPy CSR the reference builds at
main.py:134-136, plus the per-graph scratch tables the hop kernels use.

HBM layout (N nodes, nnz directed edges):
  rowptr  int64[N+1]   (int64: ogbn-papers100M symmetrised has 3.2e9 edges)
  col     int32[nnz]   ascending inside each row, duplicates removed (SciPy constructor semantics)
  bits    u64[ceil(N/64)]                          frontier bitmap (zero at rest)
  prev_bits u64[ceil(N/64)]                          membership of `previous_nodes` (zero at rest)
  node_map int32[N]    TensorMap table (modules/utils.py:112; uninitialised like the reference's)
  mult     int32[N]    column multiplicities for slice_adjacency (zero at rest)
  ind_code int32[N]    (epoch << 8 | indicator bits) replacing the N x (hops+1) indicator matrix
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


class EpochSpace:
    """Indicator epochs of one graph's `ind_code` table (epoch << 8 | indicator bits; a node's bits count only while its
    stored epoch equals the current one, which replaces the per-batch zero_() of main.py:167).  Every user of the table
    draws its epochs from HERE so that two users never tag with the same value: captured steps advance the shared DEVICE
    counter `epoch_counter()` (values 1 .. 2^23 - 1), host-driven users (eager steps, mini-batch evaluation) take
    `next_epoch()` (values 2^23 .. 2^24 - 1).  A range that runs out clears the table and starts again."""

    _HOST0 = 1 << 23

    def epoch_counter(self) -> torch.Tensor:
        t = getattr(self, "_epoch_dev", None)
        if t is None:
            t = self._epoch_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._epoch_dev_used = 0
        return t

    def note_device_epochs(self, k: int = 1):
        """Host-side count of the epochs the device counter has handed out (called once per enqueued captured step, AFTER it: the
        refill is then stream-ordered behind that step — right for steps whose step_begin is their own first launch)."""
        self.epoch_counter()
        self._epoch_dev_used += k
        if self._epoch_dev_used >= self._HOST0 - 2:
            self.ind_code.zero_(); self._epoch_dev.zero_(); self._epoch_dev_used = 0

    def would_refill(self, k: int = 1) -> bool:
        """True when reserving k more device epochs has to clear the table first."""
        self.epoch_counter()
        return self._epoch_dev_used + k >= self._HOST0 - 2

    def reserve_device_epochs(self, k: int = 1):
        """Counts k epochs BEFORE the launches that consume them are enqueued; a range that would run out is refilled NOW, i.e.
        stream-ordered in front of those launches.  For a step_begin that rides inside another step's graph (the prelude pipeline):
        refilling after that graph has been enqueued would clear the indicator table between a step's prelude and its main part
        (ADVICE r04).  The caller guarantees that nothing enqueued LATER than this call still needs the old table — in the pipeline
        that is the point right before the graph that carries this set's prelude, when this set's previous step is complete."""
        self.epoch_counter()
        if self._epoch_dev_used + k >= self._HOST0 - 2:
            self.ind_code.zero_(); self._epoch_dev.zero_(); self._epoch_dev_used = 0
        self._epoch_dev_used += k

    def next_epoch(self) -> int:
        e = getattr(self, "_epoch_host", self._HOST0 - 1) + 1
        if e >= (1 << 24):
            self.ind_code.zero_()
            if getattr(self, "_epoch_dev", None) is not None:
                self._epoch_dev.zero_(); self._epoch_dev_used = 0
            e = self._HOST0
        self._epoch_host = e
        return e


class DeviceGraph(EpochSpace):
    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, num_nodes: int):
        if not rowptr.is_cuda or not col.is_cuda:
            raise _lib.GrapesHipError("DeviceGraph lives in HBM: rowptr/col must be cuda tensors")
        assert rowptr.dtype == torch.int64 and col.dtype == torch.int32
        assert rowptr.numel() == num_nodes + 1
        if num_nodes >= 2 ** 31 - 64:
            raise ValueError("node ids are int32 on the device")
        self.rowptr = rowptr.contiguous()
        self.col = col.contiguous()
        self.num_nodes = int(num_nodes)
        self.device = rowptr.device
        dev = self.device
        W = (self.num_nodes + 63) // 64
        W1 = (W + 63) // 64
        self.bits = torch.zeros(W, dtype=torch.int64, device=dev)
        self.bits1 = None        # summary level of earlier versions: the compaction streams the level-0 words directly
        self.prev_bits = torch.zeros(W, dtype=torch.int64, device=dev)
        self.prev_bits_b = torch.zeros(W, dtype=torch.int64, device=dev)     # the captured step alternates the two per hop
        self.node_map = torch.empty(self.num_nodes, dtype=torch.int32, device=dev)
        self.mult = torch.zeros(self.num_nodes, dtype=torch.int32, device=dev)
        self.ind_code = torch.zeros(self.num_nodes, dtype=torch.int32, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self._max_degree = None

    @property
    def nnz(self) -> int:
        return self.col.numel()

    def hop_counters(self):
        """Counter tables of the counted hop build (ops.HopCounters: 4 N + N / 32 words, zero at rest), made on first use."""
        if getattr(self, "_hop_counters", None) is None:
            from . import ops
            self._hop_counters = ops.HopCounters(self.num_nodes, self.device)
        return self._hop_counters

    @property
    def max_degree(self) -> int:
        if self._max_degree is None:
            self._max_degree = int((self.rowptr[1:] - self.rowptr[:-1]).max().item()) if self.num_nodes else 0
        return self._max_degree

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_csr(cls, indptr, indices, device="cuda"):
        indptr = torch.as_tensor(np.asarray(indptr), dtype=torch.int64)
        indices = torch.as_tensor(np.asarray(indices).astype(np.int32, copy=False))
        return cls(indptr.to(device), indices.to(device), indptr.numel() - 1)

    @classmethod
    def from_scipy(cls, adjacency, device="cuda"):
        """adjacency: the scipy.sparse.csr_matrix of main.py:134-136 (already dedup'd + sorted)."""
        a = adjacency.tocsr()
        a.sort_indices()
        if a.shape[0] != a.shape[1]:
            raise ValueError("adjacency must be square")
        return cls.from_csr(a.indptr, a.indices, device)

    @classmethod
    def from_edge_index(cls, edge_index: torch.Tensor, num_nodes: int, device="cuda"):
        """Same result as sp.csr_matrix((ones(E,bool), edge_index), (N,N)) (main.py:134-136): duplicates collapse, columns
        ascending, self-loops stay.  Built on the device by the library's CSR builder (csrc/ingest_kernels.hip: counting
        placement + per-row sort / de-duplication in LDS) — one host read (the entry count) at the end."""
        from . import ops
        rowptr, col = ops.csr_build(edge_index.to(device=device, dtype=torch.int64), num_nodes)
        return cls(rowptr, col, num_nodes)

    def gcn_prepared(self):
        """PreparedGraph of the WHOLE adjacency for full-batch message passing (eval.py:47-70): self-loops are
        stripped once (PyG replaces them by the unit loop), the CSR by target is the transpose — equal to the
        CSR itself for a symmetric graph, otherwise built once with a device sort.  Cached."""
        if getattr(self, "_prepared", None) is not None:
            return self._prepared
        from . import ops
        if self.nnz >= 2 ** 31 - 1:
            raise ValueError("full-graph GCN needs an int32 edge count")
        N, dev = self.num_nodes, self.device
        row = torch.repeat_interleave(torch.arange(N, device=dev, dtype=torch.int32),
                                      (self.rowptr[1:] - self.rowptr[:-1]))
        keep = row != self.col
        row, col = row[keep], self.col[keep]
        del keep

        def to_csr(r, c):
            cnt = torch.bincount(r.long(), minlength=N)
            rp = torch.zeros(N + 1, dtype=torch.int64, device=dev)
            torch.cumsum(cnt, 0, out=rp[1:])
            return rp.to(torch.int32), c.contiguous()

        rp_s, c_s = to_csr(row, col)                       # by source (rows as stored)
        key = col.long() * N + row.long()                  # transpose: sort by (col, row)
        order = torch.argsort(key)
        t_row, t_col = col[order], row[order]
        del key, order
        symmetric = bool(torch.equal(t_row, row) and torch.equal(t_col, col))
        if symmetric:
            self._prepared = ops.PreparedGraph.from_csr(rp_s, c_s, N)
        else:
            rp_t, c_t = to_csr(t_row, t_col)
            self._prepared = ops.PreparedGraph.from_csr(rp_t, c_t, N, rp_s, c_s)
        return self._prepared

    def check_status(self, what: str = "hop pipeline"):
        """Host-side check of the device status word (synchronises).  Raises on overflow/bad ids."""
        s = int(self.status.item())
        if s:
            self.status.zero_()
            # a truncated hop may have left marks behind: restore the "zero at rest" invariant
            self.bits.zero_(); self.prev_bits.zero_(); self.prev_bits_b.zero_(); self.mult.zero_()
            bits = [n for b, n in ((1, "edge buffer overflow"), (2, "node buffer overflow"), (4, "index out of range"),
                                      (8, "one-launch scan timed out"))
                    if s & b]
            raise _lib.GrapesHipError(f"{what}: " + ", ".join(bits))


_GRAPH_CACHE: "dict[int, tuple]" = {}


def as_device_graph(adjacency, device="cuda") -> DeviceGraph:
    """Accepts a DeviceGraph or the reference's scipy CSR (uploaded once, cached by identity)."""
    if isinstance(adjacency, DeviceGraph):
        return adjacency
    key = id(adjacency)
    hit = _GRAPH_CACHE.get(key)
    if hit is not None and hit[0] is adjacency:
        return hit[1]
    g = DeviceGraph.from_scipy(adjacency, device)
    if len(_GRAPH_CACHE) > 8:
        _GRAPH_CACHE.clear()
    _GRAPH_CACHE[key] = (adjacency, g)
    return g
