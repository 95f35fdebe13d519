"""Drop-in replacements for the hot-path helpers of the reference's modules/utils.py
(same names, argument meaning, return structure and error behaviour), running on gfx950.

`adjacency` may be the SciPy CSR the reference builds (main.py:134-136; uploaded to HBM once and
cached) or a grapes_amd.graph.DeviceGraph.  Index tensors may be int64 (as in the reference) on
the CPU or on the device; results come back with the dtype/device convention of the reference
(int64, on the device of the input index tensor).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
from torch import Tensor

from .. import ops
from ..graph import DeviceGraph, as_device_graph


def _dev_i32(t: Tensor, device) -> Tensor:
    return t.to(device=device, dtype=torch.int32, non_blocking=True).contiguous()


class _SampleFn(torch.autograd.Function):
    """log_prob = Bernoulli(logits).log_prob(mask) with the draw fused in (utils.py:37-71)."""

    @staticmethod
    def forward(ctx, logits_flat, k, uniforms, candidate_ids, philox):
        res = ops.gumbel_topk(logits_flat, k, uniforms=uniforms, candidate_ids=candidate_ids,
                              philox_seed=philox[0] if philox else 0, philox_offset=philox[1] if philox else 0,
                              want_keys=False, want_stats=True)
        ctx.save_for_backward(logits_flat, res["mask"])
        ctx.aux = res
        log_prob = res["log_prob"]
        ctx.mark_non_differentiable(res["mask"], res["kept_pos"], res["stats"])
        return log_prob, res["mask"], res["kept_pos"], res["stats"]

    @staticmethod
    def backward(ctx, g_lp, g_mask, g_pos, g_stats):
        logits_flat, mask = ctx.saved_tensors
        dl = ops.bernoulli_logprob_bwd(logits_flat, mask, grad_vec=g_lp.contiguous())
        return dl, None, None, None, None


def sample_neighborhoods_from_probs(logits: Tensor, neighbor_nodes: Tensor, num_samples: int = -1,
                                    uniforms: Tensor = None, philox: Tuple[int, int] = None
                                    ) -> Tuple[Tensor, Tensor, Dict[str, Tensor]]:
    """modules/utils.py:13-71.  Exact-k without-replacement draw by Gumbel-top-k.

    Extra (optional) arguments: ``uniforms`` — the torch.rand(n) values to use for the Gumbel noise
    (default: one torch.rand(n) on logits.device, exactly what the reference's
    ``Gumbel(0,1).sample((n,))`` consumes); ``philox=(seed, offset)`` — generate them in-kernel."""
    k = num_samples
    n = neighbor_nodes.shape[0]
    if not logits.is_cuda:
        raise ops._lib.GrapesHipError("logits must be a cuda tensor (grapes_amd has no CPU path)")
    flat = logits.reshape(-1).contiguous()
    if flat.dtype != torch.float32:
        flat = flat.float()
    if k < n:
        assert k > 0                                                    # utils.py:35
        if uniforms is None and philox is None:
            uniforms = torch.rand(n, device=logits.device)              # utils.py:40 (one torch.rand(n))
    else:
        k = max(n, 1)
        uniforms = None
    cand = _dev_i32(neighbor_nodes, logits.device) if neighbor_nodes.is_cuda else None
    log_prob, mask, kept_pos, stats = _SampleFn.apply(flat, k, uniforms, cand, philox)
    if num_samples >= n:                                                # utils.py:31-33
        return neighbor_nodes, log_prob, {}
    if neighbor_nodes.is_cuda:
        kept = neighbor_nodes[kept_pos.long()]
    else:
        kept = neighbor_nodes[kept_pos.cpu().long()]                    # utils.py:60 (D2H, as the reference)
    stats_dict = {"min_prob": stats[0], "max_prob": stats[1], "mean_entropy": stats[2], "std_entropy": stats[3]}
    return kept, log_prob, stats_dict


def get_neighborhoods(nodes: Tensor, adjacency) -> Tensor:
    """modules/utils.py:74-82.  int64[2,e] on nodes.device: (queried node, neighbour), query order
    then ascending column."""
    g: DeviceGraph = as_device_graph(adjacency)
    nd = _dev_i32(nodes, g.device)
    eoff, d_e = ops.frontier_offsets(g.rowptr, nd)
    e = int(d_e.item())                      # the API returns an exactly sized tensor -> one sync
    if e == 0:
        return torch.zeros((2, 0), dtype=torch.long, device=nodes.device)
    src, dst, _ = ops.frontier_expand(g.rowptr, g.col, nd, eoff, e)
    return torch.stack([src, dst], dim=0).to(device=nodes.device, dtype=torch.long)


_SLICE_SCRATCH = {}       # per device: capacity of the expansion / survivor buffers of slice_adjacency (grows, never shrinks)


def slice_adjacency(adjacency, rows: Tensor, cols: Tensor) -> Tensor:
    """modules/utils.py:85-95.  Edges of A[rows][:, cols] as global-id pairs.  ONE host read per call: the expansion and the
    filter run inside a capacity with their sizes on the device (edge count, survivor count, status word), which are read
    together at the end; the capacity doubles and the call repeats in the rare case that it was too small (a first call, a
    batch of hub rows, duplicate column ids multiplying edges)."""
    g: DeviceGraph = as_device_graph(adjacency)
    r = _dev_i32(rows, g.device)
    c = _dev_i32(cols, g.device)
    if r.numel() == 0 or c.numel() == 0:
        return torch.zeros((2, 0), dtype=torch.long, device=rows.device)
    ops.slice_mark(g.mult, c)
    eoff, d_e = ops.frontier_offsets(g.rowptr, r)
    cap = max(_SLICE_SCRATCH.get(g.device, 1 << 16), 1)
    while True:
        src, dst, _ = ops.frontier_expand(g.rowptr, g.col, r, eoff, cap, status=g.status)
        osrc, odst, cnt = ops.slice_filter(g.mult, src, dst, cap, d_e=d_e, status=g.status)
        e, m, st = torch.cat([d_e, cnt, g.status]).tolist()             # the call's one host read
        if st == 0 and e <= cap:
            break
        g.status.zero_()
        cap = max(2 * cap, 2 * int(e))
    _SLICE_SCRATCH[g.device] = cap
    ops.slice_mark(g.mult, c, unmark=True)
    if m == 0:
        return torch.zeros((2, 0), dtype=torch.long, device=rows.device)
    return torch.stack([osrc[:m], odst[:m]], dim=0).to(device=rows.device, dtype=torch.long)


class TensorMap:
    """modules/utils.py:98-120 on the device.  ``map_tensor`` is int32 in HBM (ids < 2^31)."""

    def __init__(self, size, device="cuda", values_device="cpu"):
        size = int(size)
        self.size = size
        self.device = torch.device(device)
        self.map_tensor = torch.empty(size, dtype=torch.int32, device=self.device)   # uninitialised, utils.py:112
        # `values` stays where the reference's is — on the host: main.py:189-190,252 index it with the CPU boolean masks and use
        # the result to index CPU tensors (indicator_features, data.x)
        self.values_device = torch.device(values_device)
        self._values = None

    @property
    def values(self) -> Tensor:
        if self._values is None:
            self._values = torch.arange(self.size, device=self.values_device)        # utils.py:113
        return self._values

    def update(self, keys: Tensor):
        ops.tensormap_update(self.map_tensor, _dev_i32(keys.reshape(-1), self.device))

    def map(self, keys: Tensor) -> Tensor:
        k = _dev_i32(keys.reshape(-1), self.device)
        out = ops.tensormap_map(self.map_tensor, k)
        return out.to(device=keys.device, dtype=keys.dtype).reshape(keys.shape)
