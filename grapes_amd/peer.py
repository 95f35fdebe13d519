"""Feature rows of a 1-D node-partitioned X read IN PLACE from the GPUs that own them (SURVEY §8e — new design, the reference
is single-device; the calls being distributed are main.py:199-204 + modules/gcn.py:32).

MI355X-first: the 8 GPUs of a node are fully connected by xGMI and every GPU can load from every other GPU's HBM once the
allocation is mapped (hipIpc memory handles — the mechanism RCCL's own intra-node transport uses).  So a hop's halo needs no
exchange protocol at all: each rank maps the other ranks' shards once, and the fused gather-SpMM (csrc/spmm_kernels.hip,
gcn_aggregate_gather_head5_k<.., PEER>) picks the shard of every row it reads from a table of (base, first row) pairs that
lives in registers.  Compared with dist.PartitionedGraph's request/reply form (all-gather of id lists + all-to-all of rows per
layer boundary: 9 collectives and 10 hipGraph segments per step) the step keeps ONE collective — the gradient all-reduce —
and stays one captured graph up to it; the bytes that cross a link are exactly the rows a hop touches.

The adjacency is replicated (ogbn-products 0.5 GB, papers100M 13.8 GB of 288 GB): get_neighborhoods stays local.

`PeerFeatures.open` is collective over the process group: every rank calls it with its own shard.  Only aggregate-first first
layers (F + indicators < hidden width: products, arxiv, papers100M — the multi-GPU configurations of BASELINE.json) read
features through it; step_graph.GraphedTrainer refuses other shapes (use dist.PartitionedGraph there).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _lib

MAX_SHARDS = 8


class PeerFeatures:
    """X [N, F] as P row shards, all of them addressable from this rank.  `bases[q]` is this process's address of shard q
    (rows [bounds[q], bounds[q+1]) at `pitch` floats per row); `local` is the shard this rank owns (a cuda tensor that keeps the
    memory alive — it must outlive every peer's mapping)."""

    is_cuda = True

    def __init__(self, local: torch.Tensor, F: int, bounds: Sequence[int], rank: int, bases: Sequence[int],
                 opened: Optional[List] = None, keep=()):
        self.local, self.F, self.pitch = local, int(F), int(local.shape[1])
        self.bounds = [int(b) for b in bounds]
        self.rank, self.P = int(rank), len(self.bounds) - 1
        if not (1 <= self.P <= MAX_SHARDS):
            raise ValueError(f"PeerFeatures: 1..{MAX_SHARDS} shards (one node), got {self.P}")
        if self.bounds[0] != 0 or any(b1 < b0 for b0, b1 in zip(self.bounds, self.bounds[1:])) or self.bounds[-1] >= 2 ** 31:
            raise ValueError("PeerFeatures: bounds must ascend from 0 and fit int32")
        self.bases = [int(b) for b in bases]
        self.shape = (self.bounds[-1], self.F)
        self.device = local.device
        self.dtype = local.dtype
        self._opened = opened or []          # (address, offset) of the mappings this object must close
        self._keep = keep                    # tensors that own shards mapped from inside this process (tests: in-process shards)
        self._c_bases = (C.c_void_p * self.P)(*self.bases)
        self._c_bounds = (C.c_int32 * (self.P + 1))(*self.bounds)

    def __del__(self):          # mappings are closed when the object goes (a rebuilt trainer must not leak them)
        try:
            self.close()
        except Exception:       # noqa: BLE001  (interpreter shutdown)
            pass

    # ------------------------------------------------------------------ construction
    @staticmethod
    def _pad(X_local: torch.Tensor):
        from . import ops
        return ops.pad_features(X_local.contiguous())

    @classmethod
    def open(cls, X_local: torch.Tensor, bounds: Sequence[int], rank: int, world: int, group=None) -> "PeerFeatures":
        """Collective: exports this rank's shard, gathers every rank's handle and maps the other shards."""
        import torch.distributed as dist
        if not X_local.is_cuda:
            raise _lib.GrapesHipError("PeerFeatures: the shard must be resident in HBM (cuda tensor)")
        local, F = cls._pad(X_local)
        # (a rank whose shard does not match its bounds must not raise ALONE in front of the collective below — the others would
        # wait in it: the verdict travels with the handle and every rank raises together)
        rows_ok = local.shape[0] == bounds[rank + 1] - bounds[rank]
        lib = _lib.load()
        handle = (C.c_ubyte * 64)()
        off = C.c_uint64(0)
        rc = lib.grapes_peer_export(C.c_void_p(local.data_ptr()), handle, C.byref(off)) if local.numel() else 0
        # (first and last row travel with the handle: every rank checks what it reads through a mapping against them)
        probe = torch.stack([local[0], local[-1]]).cpu() if local.shape[0] else torch.zeros((2, local.shape[1]))
        mine = dict(rank=rank, rc=int(rc), rows_ok=bool(rows_ok), handle=bytes(handle), offset=int(off.value), pitch=int(local.shape[1]),
                    rows=int(local.shape[0]), device=int(local.device.index or 0), probe=probe)
        if world == 1:
            infos = [mine]
        else:
            infos = [None] * world
            dist.all_gather_object(infos, mine, group=group)
        wrong = [i["rank"] for i in infos if not i["rows_ok"]]
        if wrong:
            raise ValueError(f"PeerFeatures: the shards of ranks {wrong} do not have the rows their bounds say")
        bad = [i["rank"] for i in infos if i["rc"] != 0]
        if bad:
            raise _lib.GrapesHipError(f"PeerFeatures: ranks {bad} could not export their shard (hipIpcGetMemHandle)")
        if any(i["pitch"] != mine["pitch"] for i in infos):
            raise ValueError("PeerFeatures: shards differ in row pitch")
        torch.cuda.synchronize(local.device)               # the shard's contents are complete before anybody reads them
        bases, opened, failed = [], [], None
        for q, info in enumerate(infos):
            if q == rank or info["rows"] == 0:
                bases.append(local.data_ptr() if q == rank else 0)
                continue
            ptr = C.c_void_p(0)
            rc = lib.grapes_peer_open(info["handle"], C.c_uint64(info["offset"]), C.byref(ptr))
            if rc != 0:
                failed = (q, rc)
                bases.append(0)
                continue
            bases.append(int(ptr.value))
            opened.append((int(ptr.value), info["offset"]))
            got = torch.empty((2, local.shape[1]), dtype=local.dtype, device=local.device)
            nb = local.shape[1] * 4
            rc1 = lib.grapes_peer_copy(C.c_void_p(got[0].data_ptr()), C.c_void_p(int(ptr.value)), C.c_size_t(nb), None)
            rc2 = lib.grapes_peer_copy(C.c_void_p(got[1].data_ptr()), C.c_void_p(int(ptr.value) + (info["rows"] - 1) * nb),
                                       C.c_size_t(nb), None)
            torch.cuda.synchronize(local.device)
            if rc1 or rc2 or not torch.equal(got.cpu(), info["probe"]):
                failed = (q, "rows read through the mapping differ from the owner's")
        # agree on the outcome: one rank falling back alone would leave the others waiting in a collective later
        ok = torch.tensor([0 if failed else 1], dtype=torch.int32,
                          device=local.device if (world > 1 and dist.get_backend(group) == "nccl") else "cpu")
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            for a, o in opened:
                lib.grapes_peer_close(C.c_void_p(a), C.c_uint64(o))
            raise _lib.GrapesHipError("PeerFeatures: a peer shard could not be mapped (hipIpcOpenMemHandle"
                                      + (f": shard {failed[0]}, error {failed[1]}" if failed else " on another rank") + ")")
        # empty shards get a valid (never dereferenced) address: the kernel's table takes no NULLs
        valid = next((b for b in bases if b), 0)
        if not valid:
            raise ValueError("PeerFeatures: every shard is empty")
        bases = [b if b else valid for b in bases]
        pf = cls(local, F, bounds, rank, bases, opened)
        if world > 1:
            dist.barrier(group=group)                      # nobody frees / reuses a shard before every mapping exists
        return pf

    @classmethod
    def from_shards(cls, shards: Sequence[torch.Tensor], rank: int = 0) -> "PeerFeatures":
        """All shards inside THIS process (tests, single-GPU measurements of the table path): no IPC involved."""
        from . import ops
        padded = [ops.pad_features(s.contiguous()) for s in shards]
        F = padded[0][1]
        bounds = [0]
        for p, _ in padded:
            bounds.append(bounds[-1] + p.shape[0])
        anyp = next(p for p, _ in padded if p.numel())
        return cls(padded[rank][0], F, bounds, rank, [p.data_ptr() if p.numel() else anyp.data_ptr() for p, _ in padded],
                   keep=tuple(p for p, _ in padded))

    def close(self):
        lib = _lib.load()
        for a, o in self._opened:
            lib.grapes_peer_close(C.c_void_p(a), C.c_uint64(o))
        self._opened = []

    # ------------------------------------------------------------------ what the ops layer needs
    def c_table(self):
        return self._c_bases, self._c_bounds, self.P

    def contiguous(self):
        return self

    def rows_for_check(self, ids: torch.Tensor) -> torch.Tensor:
        """X[ids] through the table (one gather-SpMM over an identity graph would do; this is the plain form for tests and the
        start-up self-check: a device-side copy per shard, not a product path)."""
        out = torch.empty((ids.numel(), self.pitch), dtype=self.dtype, device=self.device)
        idc = ids.to("cpu", torch.int64)
        import numpy as np
        own = np.searchsorted(np.asarray(self.bounds[1:]), idc.numpy(), side="right")
        for i, (v, q) in enumerate(zip(idc.tolist(), own.tolist())):
            src = self.bases[q] + (v - self.bounds[q]) * self.pitch * 4
            _hip_memcpy_d2d(out[i].data_ptr(), src, self.pitch * 4)
        torch.cuda.synchronize(self.device)
        return out[:, :self.F]


def _hip_memcpy_d2d(dst: int, src: int, nbytes: int):
    _lib.check(_lib.load().grapes_peer_copy(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(nbytes), None), "peer_copy")
