"""Drop-in ``GCN`` (reference modules/gcn.py:9-42) whose layers run the gfx950 GCNConv kernels.

state_dict keys match PyG's GCNConv inside the reference module: ``gcn_layers.{i}.lin.weight``
([out,in]) and ``gcn_layers.{i}.bias``.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Union

import torch

from .._lib import diag_switch as _sw      # A/B switches: the default unless GRAPES_DIAG=1
import torch.nn as nn
import torch.nn.functional as F

from .. import ops

_PREP_CACHE: "OrderedDict[int, tuple]" = OrderedDict()
_PREP_CACHE_SIZE = 16
# floats: full-batch inference pads a transform-first layer's output rows to a multiple of this (A/B: GRAPES_EVAL_ROW_PAD=4 is
# the default, 16-byte rows; 32 = whole 128-byte lines — measured no faster once the rows are pre-scaled: profiles/r03_bench_eval_fullbatch.json)
import os as _os
_EVAL_ROW_PAD = int(_sw("GRAPES_EVAL_ROW_PAD", "4"))            # inference: output rows padded to a multiple of this many floats (16 bytes)
_EVAL_PRESCALED = _sw("GRAPES_EVAL_PRESCALED", "1") != "0"     # A/B: 0 = per-edge dinv gather (round-2 form)


def prepare_edges(edge_index, n: int) -> ops.PreparedGraph:
    """gcn_norm + CSRs for one edge list, cached on the identity (and version) of the tensor so the
    two layers of gcn_gf and gcn_z (main.py:210,227) share one preparation."""
    if isinstance(edge_index, ops.PreparedGraph):
        return edge_index
    if hasattr(edge_index, "gcn_prepared"):        # graph.DeviceGraph: full-batch message passing (eval.py:50)
        return edge_index.gcn_prepared()
    key = id(edge_index)
    hit = _PREP_CACHE.get(key)
    if hit is not None and hit[0] is edge_index and hit[1] == edge_index._version and hit[2] == n:
        _PREP_CACHE.move_to_end(key)
        return hit[3]
    if not edge_index.is_cuda:
        raise ops._lib.GrapesHipError("edge_index must be a cuda tensor (grapes_amd has no CPU path)")
    ei = edge_index.to(torch.int32)
    prep = ops.PreparedGraph(ei[0].contiguous(), ei[1].contiguous(), n)
    _PREP_CACHE[key] = (edge_index, edge_index._version, n, prep)   # holds the tensor: its address cannot be reused
    while len(_PREP_CACHE) > _PREP_CACHE_SIZE:
        _PREP_CACHE.popitem(last=False)
    return prep


def clear_prepare_cache():
    _PREP_CACHE.clear()


class _GCNConvFn(torch.autograd.Function):
    """out = Â (X Wᵀ) + b (PyG order).  When the input needs no gradient and F_in < F_out the same value is
    computed aggregate-first, act((Â X) Wᵀ + b): the SpMM runs on the narrow side, bias/ReLU ride in the GEMM
    epilogue and the backward is a single GEMM (no transposed SpMM) — equal up to fp32 rounding."""

    @staticmethod
    def forward(ctx, x, weight, bias, prep, relu):
        f_out, f_in = weight.shape
        ctx.prep, ctx.relu = prep, relu
        ctx.agg_first = (not ctx.needs_input_grad[0]) and f_in < f_out and f_out > 1
        if ctx.agg_first:
            ax = ops.gcn_aggregate_fwd(x, prep, None, False)               # Â X        (gather-SpMM, narrow rows)
            out = ops.linear_bias_act_fwd(ax, weight, bias, relu, d_n=prep.d_n)   # (ÂX) Wᵀ + b, ReLU  (MFMA)
            ctx.save_for_backward(ax, weight, out if relu else None)
            return out
        h = ops.linear_fwd(x, weight, d_n=prep.d_n)                        # H = X W^T   (MFMA fp32)
        out = ops.gcn_aggregate_fwd(h, prep, bias, relu)                   # gather-SpMM + bias (+ReLU)
        ctx.save_for_backward(x, weight, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, out = ctx.saved_tensors
        prep = ctx.prep
        if ctx.agg_first:
            dw, dbias = ops.linear_bwd_weight_gated(dout.contiguous(), x, gate=out if ctx.relu else None, d_n=prep.d_n)
            return None, dw, dbias, None, None
        dh, dbias = ops.gcn_aggregate_bwd(dout.contiguous(), prep, relu_out=out if ctx.relu else None)
        dw = ops.linear_bwd_weight(dh, x, d_n=prep.d_n)
        dx = ops.linear_bwd_input(dh, weight, d_n=prep.d_n) if ctx.needs_input_grad[0] else None
        return dx, dw, dbias, None, None


class GCNConv(nn.Module):
    """out = D^-1/2 (A + I) D^-1/2 · X Wᵀ + b with PyG's conventions."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        a = math.sqrt(6.0 / (self.in_channels + self.out_channels))    # PyG glorot
        with torch.no_grad():
            self.lin.weight.uniform_(-a, a)
            self.bias.zero_()

    def forward(self, x, edge_index, relu: bool = False):
        if not x.is_cuda:
            raise ops._lib.GrapesHipError("GCNConv input must be a cuda tensor (grapes_amd has no CPU path)")
        x = x.contiguous()
        if x.dtype != torch.float32:
            x = x.float()
        prep = prepare_edges(edge_index, x.shape[0])
        fo, fi = self.lin.weight.shape
        if (not torch.is_grad_enabled()) and _EVAL_PRESCALED and prep.n > ops._SMALL_GRAPH and prep.items_fwd and fo > 1:
            return self._forward_full_batch_inference(x, prep, relu)
        if ((not torch.is_grad_enabled()) and not _EVAL_PRESCALED and fo % _EVAL_ROW_PAD and fo > 16 and fi >= fo and
                prep.n > ops._SMALL_GRAPH and prep.items_fwd):
            # (A/B form of the diagnostic session only, GRAPES_EVAL_PRESCALED=0: the round-2 inference path — per-edge dinv
            # gather — with the output rows padded to a multiple of GRAPES_EVAL_ROW_PAD floats; the default path above does
            # the same padding inside _forward_full_batch_inference)
            fp = (fo + _EVAL_ROW_PAD - 1) // _EVAL_ROW_PAD * _EVAL_ROW_PAD
            wp = torch.zeros((fp, fi), dtype=x.dtype, device=x.device); wp[:fo] = self.lin.weight
            bp = torch.zeros(fp, dtype=x.dtype, device=x.device); bp[:fo] = self.bias
            h = ops.linear_fwd(x, wp, d_n=prep.d_n)
            return ops.gcn_aggregate_fwd(h, prep, bp, relu)[:, :fo]
        return _GCNConvFn.apply(x, self.lin.weight, self.bias, prep, relu)


def _full_batch_inference(self, x, prep, relu):
    """Full-batch message passing without autograd (eval.py:47-70: one pass over the whole adjacency; N1).  Two changes
    against the training form, both about memory requests per aggregated edge, neither about what is computed:
      * the aggregated rows are PRE-SCALED by their own dinv (ops.scale_rows — one streaming pass), so the gather-SpMM needs
        no random 4-byte gather of dinv[source] per edge;
      * a transform-first layer whose width is not a multiple of 4 floats computes on a weight / bias zero-padded to the next
        multiple of _EVAL_ROW_PAD = 4 (ogbn-products' 47 classes -> 48 columns, a 192-byte pitch): the gathered rows are
        16-byte aligned (dwordx4 loads instead of scalar ones) and the leading columns are returned as a view.  (Padding to
        whole 128-byte lines — 64 columns, GRAPES_EVAL_ROW_PAD=32 in a diagnostic session — measured the SAME time, 6.95 ms at
        F = 48: the pass is bound by requests per edge, not by the lines a row straddles; profiles/r03_bench_eval_fullbatch.json
        was measured with the default, 48 columns.)"""
    w, b = self.lin.weight, self.bias
    fo, fi = w.shape
    if fi < fo and fi % 4 == 0 and fi > 16:                                 # aggregate on the narrow side (as _GCNConvFn)
        ax = ops.gcn_aggregate_fwd_prescaled(ops.scale_rows(x, prep.dinv), prep, None, False)
        return ops.linear_bias_act_fwd(ax, w, b, relu, d_n=prep.d_n)
    fp = (fo + _EVAL_ROW_PAD - 1) // _EVAL_ROW_PAD * _EVAL_ROW_PAD if fo > 16 else fo
    if fp <= 16 or fp % 4:
        return _GCNConvFn.apply(x, w, b, prep, relu)
    if fp != fo:
        wp = torch.zeros((fp, fi), dtype=x.dtype, device=x.device); wp[:fo] = w
        bp = torch.zeros(fp, dtype=x.dtype, device=x.device); bp[:fo] = b
    else:
        wp, bp = w, b
    h = ops.linear_fwd_row_scaled(x, wp, prep.dinv, d_n=prep.d_n)       # dinv scaling in the GEMM's epilogue (no scale_rows pass)
    out = ops.gcn_aggregate_fwd_prescaled(h, prep, bp, relu)
    return out[:, :fo] if fp != fo else out


GCNConv._forward_full_batch_inference = _full_batch_inference


class _PhiloxDropoutFn(torch.autograd.Function):
    """F.dropout on the sampler's Philox stream (ops.dropout_fwd): the mask is a function of (seed, offset, element index), so
    a captured step and an eager step draw the same one."""

    @staticmethod
    def forward(ctx, x, p, seed, offset):
        y, keep = ops.dropout_fwd(x.contiguous(), p, philox_seed=seed, philox_offset=offset)
        ctx.save_for_backward(keep)
        ctx.p = p
        return y

    @staticmethod
    def backward(ctx, dy):
        (keep,) = ctx.saved_tensors
        return ops.dropout_bwd(dy.contiguous(), keep, ctx.p), None, None, None


def _memory_allocated_mb() -> float:
    """torch.cuda.memory_allocated() / 2^20 (gcn.py:40: the second element of GCN.forward's result) read from the allocator's
    nested statistics directly — torch.cuda.memory_allocated() flattens the whole statistics dictionary first (0.14 ms per call,
    five calls per training step: profiles/eager_profile.py)."""
    try:
        st = torch._C._cuda_memoryStats(torch._C._cuda_getDevice())
        return st["allocated_bytes"]["all"]["current"] / (1024 * 1024)
    except (AttributeError, KeyError, RuntimeError):
        return torch.cuda.memory_allocated() / (1024 * 1024)


class GCN(nn.Module):
    def __init__(self, in_features: int, hidden_dims: "list[int]", dropout: float = 0.):
        super(GCN, self).__init__()
        self.dropout = dropout
        # optional: a callable  n_elements -> (seed, offset)  that hands out Philox counters (step.GrapesTrainer sets it when it
        # was given a philox_seed); without it dropout draws from torch's generator exactly like the reference
        self.philox_dropout = None
        dims = [in_features] + hidden_dims
        gcn_layers = []
        for i in range(len(hidden_dims) - 1):
            gcn_layers.append(GCNConv(in_channels=dims[i], out_channels=dims[i + 1]))
        gcn_layers.append(GCNConv(in_channels=dims[-2], out_channels=dims[-1]))
        self.gcn_layers = nn.ModuleList(gcn_layers)

    def _drop(self, x):
        # (only under autograd, i.e. inside a training step: evaluate() runs under no_grad with the module still in training mode
        # — the reference never calls .eval() — and must not consume the training step's Philox counters)
        if (self.philox_dropout is not None and self.training and self.dropout > 0.0 and x.is_cuda and
                torch.is_grad_enabled()):
            seed, offset = self.philox_dropout(x.numel())
            return _PhiloxDropoutFn.apply(x, float(self.dropout), seed, offset)
        return F.dropout(x, p=self.dropout, training=self.training)

    def forward(self, x: torch.Tensor, edge_index: Union[torch.Tensor, "list[torch.Tensor]"]):
        layerwise_adjacency = type(edge_index) == list
        # (the reference slices the ModuleList, gcn.py:31: a slice builds a NEW ModuleList on every call — add_module and its
        # hasattr probes, ~0.1 ms per layer of host time; iterate by index instead)
        n_layers = len(self.gcn_layers)
        for i in range(1, n_layers):
            layer = self.gcn_layers[i - 1]
            edges = edge_index[-i] if layerwise_adjacency else edge_index      # gcn.py:31
            x = layer(x, edges, relu=True)                                     # gcn.py:32 (ReLU fused)
            x = self._drop(x)                                                  # gcn.py:33
        edges = edge_index[0] if layerwise_adjacency else edge_index           # gcn.py:35
        logits = self.gcn_layers[n_layers - 1](x, edges)
        logits = self._drop(logits)                                            # gcn.py:37
        memory_alloc = _memory_allocated_mb()                                  # gcn.py:40
        return logits, memory_alloc
