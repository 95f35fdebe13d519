"""A8, device-driven: the same training iteration as step.GrapesTrainer (reference main.py:157-291),
but with NO host round-trip inside the step — every size stays on the device next to a fixed
capacity (`d_*` convention of include/grapes_hip.h), the backward pass is scheduled explicitly
(no autograd graph walk), and the whole iteration — three sampling hops, log-Z net, classifier,
both losses, both backward passes, both Adam updates — is captured once as ONE hipGraph and
replayed per mini-batch.  ("a whole step is one hipGraph".)  Over a dist.PartitionedGraph
(1-D node partition) the same body is captured as hipGraph segments with the RCCL
collectives launched between them (capture.SegmentedGraph) — still no host read.

Semantics are those of GrapesTrainer (which stays the readable, exact-size reference; the two are
compared step for step in tests/test_hip_parity.py): same kernels, same summation orders, same
Philox stream.  `random_sampling=True` (the reference's configs/random/*, main.py:206-207,223,272) captures the
shorter step of that mode: uniform exact-k draws, no sampler net, no log-Z net, classifier update only.  `reg_param` adds
the logit-variance regulariser of main.py:260-261 (two small launches); the classifier's `dropout` (modules/gcn.py:33,37)
draws its masks from the same Philox stream as the sampler (two small launches per layer output).

Capacities: every hop may expand up to `e_cap` edges and touch up to `e_cap + B + K` nodes; if a
batch exceeds them the kernels drop the excess and raise the device status word, which
`check()` turns into an exception — nothing is silently truncated.
"""
from __future__ import annotations

import contextlib
from typing import Dict, List, Optional

import os

import torch

from ._lib import diag_switch as _sw      # A/B switches: the default unless GRAPES_DIAG=1
import torch.nn as nn

from . import ops
from .capture import SegmentedGraph
from .graph import DeviceGraph


class _FirstLayer:
    """The first GCNConv of a model, applied to data rows  feat(v) = [X[v] | indicators(v)]  of ANY width (the reference
    builds F + num_indicators input columns, main.py:111-113: 104 on products, 131 on arxiv, 605 on Reddit, 1436 on Cora).
    The kernels work on rows of Kp = ceil4(F + num_ind) floats, so the layer keeps a zero-padded IMAGE of its weight
    [out, Kp] (refreshed from the parameter at the top of every step: one strided copy) and a padded gradient buffer that is
    copied back into `.grad` once the step's backward passes are done; when F + num_ind already is a multiple of 4 the image
    IS the parameter.  Order of operations:
      * K < out_channels  (products 104 -> 256, arxiv 131 -> 256): aggregate first — gather-SpMM on Kp-wide rows fused with
        the feature gather, then the GEMM;
      * K >= out_channels (Reddit 605 -> 256, Cora 1436 -> 256): the reference order, transform then aggregate — the GEMM
        reads feat(ids) through the id list (ops.linear_fwd_gathered), the aggregation runs on out_channels-wide rows."""

    def __init__(self, conv, F, num_ind, legacy=False):
        self.conv, self.F, self.num_ind = conv, F, num_ind
        self.K = F + num_ind
        self.Kp = self.K if legacy else (self.K + 3) // 4 * 4
        # (the gathered-operand GEMMs of the transform-first order need 16-byte rows of the layer's output: any other width —
        # a hidden_dim that is not a multiple of 4, a one-layer classifier with an odd class count — aggregates first)
        self.agg_first = legacy or self.K < conv.out_channels or conv.out_channels % 4 != 0
        self.padded = self.Kp != self.K
        w = conv.lin.weight
        if self.padded:
            self.W = torch.zeros((w.shape[0], self.Kp), dtype=w.dtype, device=w.device)
            self.dW = torch.zeros_like(self.W)
        # transform-first layers on the bf16 matrix pipe: the forward GEMM takes W as a split image (one launch per step)
        self.split = (not self.agg_first) and (not legacy) and ops.split_gathered_available(w.shape[0])
        self.image = torch.empty(int(ops.lib().grapes_weight_split_image_bytes(self.K)), dtype=torch.uint8,
                                 device=w.device) if self.split else None

    def refresh(self):
        if self.split:      # (the padded fp32 copy comes out of the image launch)
            ops.weight_split_image(self.conv.lin.weight.detach(), self.image, w_pad=self.W if self.padded else None)
        elif self.padded:
            self.W[:, :self.K].copy_(self.conv.lin.weight.detach())

    @property
    def weight(self):
        return self.W if self.padded else self.conv.lin.weight

    @property
    def grad(self):
        # (split layers: the dW kernel's slab sum writes the parameter's own [out, K] gradient — no padded buffer, no copy)
        return self.dW if (self.padded and not self.split) else self.conv.lin.weight.grad

    @property
    def grad_direct(self):
        """The parameter's own gradient for a backward launch that writes its [out, K] layout itself (the gate-bit slab sums):
        marks the padded buffer as unused, so that no strided copy publishes it."""
        self.direct = True
        return self.conv.lin.weight.grad

    def publish_grad(self):
        if self.padded and not self.split and not getattr(self, "direct", False):
            self.conv.lin.weight.grad.copy_(self.dW[:, :self.K])

    @property
    def publishes(self):          # a strided copy of the padded gradient follows the backward passes
        return self.padded and not self.split and not getattr(self, "direct", False)


class _StepSet:
    """What ONE step in flight owns besides the models (GraphedTrainer.attach_loader): the prelude pipeline alternates two."""

    def __init__(self, g, targets, epoch_t, ctr):
        self.g, self.targets, self.epoch_t, self.ctr = g, targets, epoch_t, ctr
        self.out: Dict[str, torch.Tensor] = {}
        self.G = None                     # capture.SegmentedGraph: this set's main part + the other set's prelude riding in it
        self.gen = None
        self.program = -1                 # rider program of this set's prelude (include/grapes_hip.h)
        self.riders = (0, 0)
        self.handoff = None


class _StepChain:
    """Handle of one grapes_graph_chain (destroyed with the trainer that built it)."""

    def __init__(self, handle, nodes: int):
        self.handle, self.nodes = handle, nodes

    def __del__(self):
        try:
            ops.lib().grapes_graph_chain_destroy(self.handle)
        except Exception:
            pass


class GraphedTrainer:
    _lanes = 0

    def __init__(self, graph: DeviceGraph, X: torch.Tensor, y: torch.Tensor, gcn_c: nn.Module, gcn_gf: nn.Module,
                 gcn_z: nn.Module, *, batch_size: int, sampling_hops: int = 2, num_samples: int = 16,
                 use_indicators: bool = True, loss_coef: float = 1e4, log_z_init: float = 0.0,
                 reinforce_baseline: bool = False, optimizer_c: Optional[torch.optim.Optimizer] = None,
                 optimizer_gf: Optional[torch.optim.Optimizer] = None, e_cap: int = 1 << 17, philox_seed: int = 0,
                 capture: bool = True, grad_sync=None, auto_calibrate: bool = True, random_sampling: bool = False,
                 reg_param: float = 0.0, pipeline: bool = True, evaluate: bool = False):
        # evaluate=True: the mini-batch EVALUATION of the reference (eval.py:71-163) as the same captured step — the sampler net's
        # greedy draws (top-k of the inclusion probabilities, eval.py:126-130, no noise), slice_adjacency with its arguments
        # swapped (rows = previous_nodes, cols = the new layer: eval.py:140-142), the classifier's forward pass and the targets'
        # predicted classes (eval.py:153-155); no log-Z net, no loss, no backward pass, no optimiser.  out["pred"] = int64[B].
        self.evaluate = bool(evaluate)
        if self.evaluate:
            gcn_z, optimizer_c, optimizer_gf, pipeline = None, None, None, False
            if random_sampling:
                raise ValueError("evaluate=True draws greedily from the sampler net")
        self.random_sampling = bool(random_sampling)
        self.reg_param = float(reg_param)                  # main.py:260-261
        self.dropout = 0.0 if evaluate else float(getattr(gcn_c, "dropout", 0.0) or 0.0)      # main.py:110: the classifier's only (eval(): none)
        if not self.random_sampling and (gcn_gf is None or (gcn_z is None and not self.evaluate)):
            raise ValueError("the sampler net and the log-Z net may only be omitted with random_sampling=True")
        if self.random_sampling:
            gcn_gf = gcn_z = None                 # main.py:206-207,223: neither net runs (nor trains) in that mode
            optimizer_gf = None
        self.partitioned = hasattr(graph, "features")      # dist.PartitionedGraph: halo features (and rows) by all-to-all
        # ... whose adjacency may be replicated (features only partitioned): expansion is then local, as on one GPU
        self.part_adj = self.partitioned and not getattr(graph, "adjacency_replicated", False)
        if X is None:
            if not self.partitioned:
                raise ValueError("X may only be omitted with a dist.PartitionedGraph (which owns its feature shard)")
        elif not X.is_cuda:
            raise ops._lib.GrapesHipError("X must be resident in HBM (cuda tensor)")
        for opt in (optimizer_c, optimizer_gf):
            if capture and opt is not None and not all(gp.get("capturable", False) for gp in opt.param_groups):
                raise ValueError("optimizers must be built with capturable=True to live inside the captured step")
        # --embed_nodes (main.py:89-100,116): X is an nn.Parameter that optimizer_c owns.  The classifier's first layer then also
        # returns its input gradient, whose rows are scattered into the dense X.grad (zeroed at the top of the step, as
        # optimizer_c.zero_grad() does) before the optimiser launch updates ALL rows (dense Adam, like the reference).  The sampler
        # / log-Z nets' input gradients are not formed: the reference never uses them (optimizer_gf does not own the embeddings
        # and optimizer_c.zero_grad() clears what loss_gfn.backward() left, main.py:263-289).
        self.embed = isinstance(X, nn.Parameter) and X.requires_grad
        if self.embed:
            if X.shape[1] % 4 != 0 or not X.is_contiguous() or hasattr(X, "c_table"):
                raise ValueError("learned node embeddings need node_emb_dim % 4 == 0 (16-byte rows: the kernels read the parameter "
                                 "in place)")
            if X.grad is None:
                X.grad = torch.zeros_like(X)
        self.g, self.X, self.y = graph, (None if X is None else (X if self.embed else X.contiguous())), y
        self.gcn_c, self.gcn_gf, self.gcn_z = gcn_c, gcn_gf, gcn_z
        self.B, self.hops, self.K = batch_size, sampling_hops, num_samples
        self.F = X.shape[1] if X is not None else graph.feature_dim
        self.num_ind = sampling_hops + 1 if use_indicators else 0            # main.py:104-107
        # partitioned features: the fused gather-SpMM reads the exchanged halo rows where they arrive (no assembled copy) when
        # the row widths need no padding
        self._halo_in_place = (self.partitioned and getattr(getattr(graph, 'ops', None), 'in_place_halo', False) and
                               self.F % 4 == 0 and (self.F + self.num_ind) % 4 == 0 and
                               _sw('GRAPES_HALO_IN_PLACE', '1') != '0')
        self.loss_coef, self.log_z_init, self.reinforce = loss_coef, log_z_init, reinforce_baseline
        self.opt_c, self.opt_gf = optimizer_c, optimizer_gf
        # capacities never need to exceed the graph itself (a small graph with the default e_cap would otherwise size — and
        # clear — every per-hop scratch for 131k edges)
        self.e_cap = int(min(int(e_cap), max(int(getattr(graph, "nnz", e_cap)), 1) + 1)) if not self.partitioned else int(e_cap)
        self._rp, self._cl = ((graph.rowptr_full, graph.col_full) if (self.partitioned and not self.part_adj)
                              else (graph.rowptr, graph.col))
        self.n_cap = self.e_cap + batch_size + num_samples + 1
        if not self.partitioned:
            self.n_cap = min(self.n_cap, graph.num_nodes + 1)
        self.nall_cap = batch_size + sampling_hops * num_samples + 1
        self.seed = int(philox_seed)
        self._halo_code = None
        self.grad_sync = grad_sync
        dev = graph.device
        self.targets = torch.zeros(batch_size, dtype=torch.int32, device=dev)          # static input
        self.epoch_t = graph.epoch_counter()         # indicator epoch (device; shared by every captured step on this graph)
        self.philox_off = torch.zeros(1, dtype=torch.int64, device=dev)                # Philox counter (device)
        # device counters of every graph build of a step in one persistent table: column 2 = edges one aggregation sums
        self._ctr = torch.zeros((2 * sampling_hops, 4), dtype=torch.int32, device=dev)
        self._loader = None                                                            # see attach_loader
        if y.dim() == 2 and y.dtype != torch.float32:
            self.y = y = y.to(torch.float32)                                           # BCEWithLogitsLoss targets (main.py:120-123)
        self._models = [m for m in (gcn_c, gcn_gf, gcn_z) if m is not None]
        if grad_sync is not None and self.embed:
            raise ValueError("learned node embeddings on several GPUs are not built (the dense N x F gradient would join the "
                             "all-reduce); use one GPU")
        if grad_sync is not None and hasattr(grad_sync, "make_bucket"):
            # the gradients live inside the all-reduce bucket from the start (same order as the sync call in _step_impl)
            grad_sync.make_bucket([p for m in self._models for p in m.parameters()])
        for m in self._models:
            for p in m.parameters():
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
        self.out: Dict[str, torch.Tensor] = {}
        # the prelude pipeline (attach_loader + _run_pipelined): a self-feeding captured step over a plain DeviceGraph
        # (not with learned embeddings: hop 0's gather-SpMM reads X, which the optimiser then changes — the prelude would not be
        # weight-independent)
        self._pipeline_ok = (bool(pipeline) and capture and not self.partitioned and isinstance(graph, DeviceGraph) and
                             not self.embed and _sw("GRAPES_PRELUDE_PIPELINE", "1") != "0")
        self._prelude_lane = 0             # (attach_loader gives a pipelined trainer a scratch lane of its own)
        self._sets = None
        self._chains: Dict = {}                            # (parity of the first step, steps) -> _StepChain (run_steps)
        self.graph_obj = None
        self._want_capture = capture
        self.steps_done = 0
        self.eager_steps = 3 if self.partitioned else 2    # warm-up (+ one step with the calibrated slot size)
        self.auto_calibrate = auto_calibrate
        self._halo = None
        self._fused_adam = None
        self._mirrored, self._mirrors_current = set(), False      # weights whose padded copy / image the optimiser launch maintains
        # resident features with 16-byte aligned rows (a zero-padded copy when F % 4 != 0), and the first layers' weight images
        # X as peer.PeerFeatures: a 1-D row partition over the GPUs of the node, every shard mapped into this process — the fused
        # gather-SpMM reads a row from the GPU that owns it (xGMI loads); the step is the single-GPU step, no exchange at all
        self.peers = X if hasattr(X, "c_table") else None
        self.Xp = None if X is None else (X if self.peers is not None else (X.detach() if self.embed else ops.pad_features(self.X)[0]))
        # (Infinity-Cache prefetch of the rows a hop will gather: local HBM only)
        self._prefetch_X = (self.Xp if self.peers is None else (self.peers.local if self.peers.P == 1 else None))
        leg = self.partitioned
        self._fl = {id(m.gcn_layers[0]): _FirstLayer(m.gcn_layers[0], self.F, ni, legacy=leg) for m, ni in
                    ((gcn_gf, self.num_ind), (gcn_z, 0), (gcn_c, 0)) if m is not None}
        # A/B form (diagnostic session only, measured SLOWER: Reddit 1.418 against 1.366 ms/step): transform-first layers on the bf16
        # matrix pipe gather rows of X through the id lists in their GEMMs — X split once into its bf16 planes (+1.5 x its bytes)
        self._planes = None
        if (self.Xp is not None and self.peers is None and not self.embed and any(fl.split for fl in self._fl.values()) and
                _sw("GRAPES_FEATURE_PLANES", "0") != "0"):
            self._planes = ops.FeaturePlanes(self.Xp)
        if self.embed and not self._fl[id(gcn_c.gcn_layers[0])].agg_first:
            raise NotImplementedError("--embed_nodes: the embedding gradient through a transform-first classifier layer "
                                      "(node_emb_dim >= hidden_dim) is not built")
        if self.peers is not None and not all(fl.agg_first for fl in self._fl.values()):
            raise ValueError("peer-mapped features are read by the aggregate-first first layers only (F + indicators < hidden "
                             "width); use dist.PartitionedGraph for this shape")
        # main.py:207: every candidate's logit is 100 under random_sampling (read through nb_local like the net's output)
        self._rnd_logits = torch.full((self.n_cap,), 100.0, dtype=torch.float32, device=dev) if self.random_sampling else None
        # The step is ONE chain of launches on one stream.  Parallel graph branches (log-Z net, per-hop sampler backward
        # passes, classifier backward on side streams) were measured SLOWER on MI355X / ROCm 7 (2.23 vs 1.69 ms/step: the
        # cross-queue dependencies of a replayed hipGraph cost more than 5-50 us kernels overlap) and shared the per-device
        # ticket / sync scratch words between concurrent kernels, so that form was removed in round 2.

    # ------------------------------------------------------------------ GCNConv, explicit fwd / bwd
    @staticmethod
    def _conv_fwd(conv, x, prep, relu):
        h = ops.linear_fwd(x, conv.lin.weight, d_n=prep.d_n)                           # H = X W^T (MFMA)
        return ops.gcn_aggregate_fwd(h, prep, conv.bias, relu)                         # Â H + b (+ReLU)

    @staticmethod
    def _conv_bwd(conv, x, out, dout, prep, relu, need_dx, accumulate, defer=None):
        dh, _ = ops.gcn_aggregate_bwd(dout, prep, relu_out=out if relu else None, dbias=conv.bias.grad,
                                      accumulate_bias=accumulate)
        if need_dx and defer is not None and _sw("GRAPES_DW_DX_PAIR", "1") != "0":
            # the weight and the input gradient read the same dh and not each other: one launch, side by side
            return ops.linear_bwd_weight_and_input(dh, x, conv.lin.weight, d_n=prep.d_n, out=conv.lin.weight.grad,
                                                   accumulate=accumulate, defer=defer)
        ops.linear_bwd_weight(dh, x, d_n=prep.d_n, out=conv.lin.weight.grad, accumulate=accumulate, defer=defer)
        return ops.linear_bwd_input(dh, conv.lin.weight, d_n=prep.d_n) if need_dx else None

    # ---- first layers (input = data rows): see _FirstLayer
    def _first_fwd(self, conv, ids, prep, num_ind, ep, halo=None, head=None, relu=True, defer_head=False, pair=None, ax=None,
                   h_pre=None, keep=None, x_rows=None):
        """-> (state, act[, head output]).  state = the aggregated input Â[X|ind] (aggregate-first: the operand of the dW
        GEMM) or the id list (transform-first: dW re-reads the rows through it).  ax: Â[X|ind] when the step's prelude has
        formed it already (it depends on the batch only, not on the weights)."""
        st = self._fl[id(conv)]
        code = self.g.ind_code if num_ind else None
        dep = ep if num_ind else None
        if not st.agg_first:                       # reference order with the gathered-operand GEMM
            h = h_pre                              # (formed together with another net's over the same rows: _dual_first_gemm)
            if h is None:
                h = ops.linear_fwd_gathered(self.Xp, self.F, ids, st.weight, code, 0, num_ind, d_epoch=dep, d_n=prep.d_n,
                                            w_image=st.image)
            if head is not None and _sw("GRAPES_FUSED_HEAD", "1") != "0":
                # + the X W step of the 1-wide layer that follows, from the rows while the aggregation holds them
                r = ops.gcn_aggregate_fwd_head(h, prep, conv.bias, relu, head.lin.weight.view(-1),
                                               # (on Cora's <= 2.7k rows the row-per-wavefront launch over the activations
                                               # was the faster one per launch — but the gate bits are what lets the hops'
                                               # backward chains and weight-gradient GEMMs run as ONE chain / ONE launch:
                                               # Cora 0.532 -> 0.493 ms/step; A/B: GRAPES_R1_BITS_MIN=16384)
                                               want_bits=(h.shape[0] >= int(_sw("GRAPES_R1_BITS_MIN", "0")) and
                                                          _sw("GRAPES_R1_BITS", "1") != "0"))
                if r is not None:
                    if len(r) > 2 and r[2] is not None:
                        r[0]._gate_bits = r[2]           # (the backward aggregation reads 32 bytes of gates per row, not the row)
                    return ids, r[0], ops.gcn_aggregate_fwd(r[1], prep, head.bias, False)  # Â (act w2ᵀ) + b2
            act = ops.gcn_aggregate_fwd(h, prep, conv.bias, relu)
            if head is not None:
                return ids, act, self._conv_fwd(head, act, prep, False)
            return ids, act
        if self.partitioned and x_rows is not None:
            # the rows were received by this step's hop fetches (all_nodes are their batch rows): no exchange (dist.rows_from_kept)
            ax = ops.gcn_aggregate_fwd(x_rows, prep, None, False)
        elif self.partitioned:
            if halo is None:
                halo = self.g.fetch_halo(ids, d_n=prep.d_n, keep=keep)                     # halo rows (all-to-all)
                self._halo = halo
            if self._halo_in_place and prep.head_ids is not None:
                # Â [X | ind] straight from the exchanged rows: the head records were built on their positions in `back`
                ax = ops.gcn_aggregate_gather(halo["back"].view(-1, self.F), prep.head_ids, prep, self._halo_code if num_ind else None,
                                              0, num_ind, d_epoch=dep, F=self.F)
            else:
                x = self.g.assemble(halo, ind_code=self.g.ind_code, d_epoch=ep, num_ind=num_ind)
                ax = ops.gcn_aggregate_fwd(x, prep, None, False)
        elif ax is None:
            ax = ops.gcn_aggregate_gather(self.Xp, ids, prep, code, 0, num_ind, d_epoch=dep, F=self.F)   # Â [X | ind | 0]
        if head is not None and _sw("GRAPES_FUSED_HEAD", "1") != "0":      # + the XW step of the 1-wide layer that follows, from the same output tiles
            if relu and self._gate_bits(ax, st, conv):
                # the head is the activations' only consumer: keep 32 bytes of ReLU gate bits per row for the backward pass
                # instead of writing (and reading back) n x H floats
                if pair is not None:
                    # ... and a second net's first layer over the same rows (pair = (st_b, conv_b, head_b): the log-Z net at hop 0,
                    # reading the leading columns of the same aggregate) rides in the same launch
                    st_b, conv_b, head_b = pair
                    xb = ax[:, :st_b.Kp]
                    if self._gate_bits(xb, st_b, conv_b) and defer_head:
                        r = ops.linear_relu_head_fwd_bits_pair(ax, st.weight, conv.bias, head.lin.weight, xb, st_b.weight,
                                                               conv_b.bias, head_b.lin.weight, d_n=prep.d_n)
                        if r is not None:
                            return ax, r[0], ("deferred", r[1]), (xb, r[2], r[3])
                act, hw = ops.linear_relu_head_fwd_bits(ax, st.weight, conv.bias, head.lin.weight, d_n=prep.d_n)
            else:
                act, hw = ops.linear_bias_act_head_fwd(ax, st.weight, conv.bias, relu, head.lin.weight, d_n=prep.d_n)
            if defer_head:            # the head's aggregation rides in the sampler's first launch (ops.gumbel_topk(agg=...))
                return ax, act, ("deferred", hw)
            return ax, act, ops.gcn_aggregate_fwd(hw, prep, head.bias, False)             # Â (act w2ᵀ) + b2
        act = ops.linear_bias_act_fwd(ax, st.weight, conv.bias, relu, d_n=prep.d_n)        # ReLU((ÂX) Wᵀ + b)
        if head is not None:
            return ax, act, self._conv_fwd(head, act, prep, False)
        return ax, act

    @staticmethod
    def _gate_bits(ax, st, conv):
        return (_sw("GRAPES_GATE_BITS", "1") != "0" and st.agg_first and
                ops.split_gemm_available(ax.shape[0], ax.shape[1], conv.out_channels) and
                tuple(st.weight.shape) == (conv.out_channels, ax.shape[1]) and st.weight.is_contiguous())

    def _first_bwd(self, conv, state, act, dact, prep, accumulate, num_ind=0, ep=None, relu=True, hop=None, defer=None):
        """Backward of a first layer given d(its output) = dact [n, out]; the input needs no gradient."""
        st = self._fl[id(conv)]
        if st.agg_first:
            ops.linear_bwd_weight_gated(dact, state, gate=act if relu else None, d_n=prep.d_n, dw=st.grad, dbias=conv.bias.grad,
                                        accumulate=accumulate, defer=defer)
            return
        dh, _ = ops.gcn_aggregate_bwd(dact, prep, relu_out=act if relu else None, dbias=conv.bias.grad,
                                      accumulate_bias=accumulate)
        ops.linear_bwd_weight_gathered(dh, self.Xp, self.F, state, st.grad, self.g.ind_code if num_ind else None, 0, num_ind,
                                       d_epoch=ep if num_ind else None, d_n=prep.d_n, accumulate=accumulate, split=st.split,
                                       # the indicator bits this hop's forward pass saw (main.py:168,191): later hops add theirs
                                       ind_mask=(((1 << (hop + 1)) - 1) | (1 << (num_ind - 1))) if (num_ind and hop is not None) else 0)

    def _r1_ok(self, conv1, act1):
        """the rank-1 gate-bit form of _head_bwd applies (transform-first layer whose forward kept gate bits)"""
        st = self._fl[id(conv1)]
        return (not st.agg_first and act1.shape[1] % 4 == 0 and 16 < act1.shape[1] <= 256 and
                _sw("GRAPES_BWD_RANK1", "1") != "0" and getattr(act1, "_gate_bits", None) is not None)

    def _head_bwd(self, conv1, conv2, ax, act1, dhead, prep, accumulate, db2_done=False, dh2=None, num_ind=0, ep=None,
                  hop=None, dh_r1=None, dw_batch=None):
        """Backward of  first layer -> ReLU -> 1-wide head  given d(head output) = dhead [n,1].  Aggregate-first layers: the
        gradient the head sends back, dAct = dh2 ⊗ w2, is rank-1 and is formed inside the dW GEMM's operand loads (with the
        ReLU mask) instead of being written out (n x H floats) and read back."""
        st = self._fl[id(conv1)]
        w1g, b1g, w2g, b2g = st.grad, conv1.bias.grad, conv2.lin.weight.grad, conv2.bias.grad
        if dh2 is not None:   # Âᵀ dhead already formed (sampler_head_bwd_multi)
            pass
        elif db2_done:        # the head's bias gradient (sum of dhead) was produced by the kernel that wrote dhead
            dh2, _ = ops.gcn_aggregate_bwd(dhead, prep, want_bias=False)
        else:
            dh2, _ = ops.gcn_aggregate_bwd(dhead, prep, dbias=b2g, accumulate_bias=accumulate)
        if (not st.agg_first and act1.shape[1] % 4 == 0 and act1.shape[1] > 16 and
                _sw("GRAPES_BWD_RANK1", "1") != "0"):
            # reference order, rank-1 upstream gradient: dW2, db1 and dH = Âᵀ((dh2 ⊗ w2) ⊙ [act > 0]) without writing the outer
            # product or its masked copy (three launches and 5 n H floats of traffic on Reddit's 77k-row frontier less)
            dh = dh_r1
            if dh is None:
                dh = ops.gcn_aggregate_bwd_rank1(act1, dh2.view(-1), conv2.lin.weight.view(-1), prep, dw_head=w2g.view(-1), dbias=b1g,
                                                 accumulate=accumulate, gate_bits=getattr(act1, "_gate_bits", None))
            mask = (((1 << (hop + 1)) - 1) | (1 << (num_ind - 1))) if (num_ind and hop is not None) else 0
            if dw_batch is not None:       # the caller issues the nets' weight-gradient GEMMs together (one launch, one slab sum)
                dw_batch.append(dict(dh=dh, ids=ax, dw=st.grad, ind_code=self.g.ind_code if num_ind else None, num_ind=num_ind,
                                     d_n=prep.d_n, accumulate=accumulate, ind_mask=mask, split=st.split, ep=ep if num_ind else None))
                return
            ops.linear_bwd_weight_gathered(dh, self.Xp, self.F, ax, st.grad, self.g.ind_code if num_ind else None, 0, num_ind,
                                           d_epoch=ep if num_ind else None, d_n=prep.d_n, accumulate=accumulate, split=st.split,
                                           ind_mask=mask)
            return
        if not st.agg_first:  # reference order: dW2 = dh2ᵀ act, dAct = dh2 ⊗ w2 written out, then the layer's own backward
            ops.linear_bwd_weight(dh2, act1, d_n=prep.d_n, out=w2g, accumulate=accumulate)
            dact = ops.linear_bwd_input(dh2, conv2.lin.weight, d_n=prep.d_n)
            self._first_bwd(conv1, ax, act1, dact, prep, accumulate, num_ind=num_ind, ep=ep, hop=hop)
            return
        if isinstance(act1, ops.GateBits):                  # forward kept the ReLU gate bits only: dW1, db1, dW2 from them
            ops.linear_bwd_weight_bits_multi([act1], [ax], [dh2.view(-1)], [prep.d_n], conv2.lin.weight.view(-1), st.weight,
                                             conv1.bias, st.grad_direct, dbias=b1g, dw_head=w2g.view(-1), accumulate=accumulate)
            return
        fi, fo = ax.shape[1], act1.shape[1]
        if ax.stride(0) != fi:                              # a leading-columns view of a wider matrix (log-Z net at hop 0)
            ops.linear_bwd_weight_gated_strided(ax, act1, dh2.view(-1), conv2.lin.weight.view(-1), w1g, dbias=b1g,
                                                dw_head=w2g.view(-1), d_n=prep.d_n, accumulate=accumulate)
        elif fi % 4 == 0 and fi % 128 != 0 and fo % 4 == 0:   # dW1, db1 and the head's dW2 = dh2ᵀ·act1 from ONE split-K GEMM
            ops.linear_bwd_weight_gated(None, ax, gate=act1, d_n=prep.d_n, dw=w1g, dbias=b1g, accumulate=accumulate,
                                        row_scale=dh2.view(-1), col_vec=conv2.lin.weight.view(-1), dw_head=w2g.view(-1))
        else:
            ops.linear_bwd_weight(dh2, act1, d_n=prep.d_n, out=w2g, accumulate=accumulate)
            dact = ops.linear_bwd_input(dh2, conv2.lin.weight, d_n=prep.d_n)
            ops.linear_bwd_weight_gated(dact, ax, gate=act1, d_n=prep.d_n, dw=w1g, dbias=b1g, accumulate=accumulate)

    def weights_changed(self):
        """Call after writing the models' weights from OUTSIDE the trainer's optimiser step (a checkpoint load, a copy_): the first
        layers' padded copies and split images are refreshed now — in steady state they follow the weights through the optimiser
        launch (ops.FusedAdam mirrors), and a captured step holds no launch that would rebuild them."""
        for fl in self._fl.values():
            fl.refresh()

    def _optim_step(self):
        """Both Adam updates in one launch when the optimisers are torch.optim.Adam with device-resident state;
        anything else steps through its own .step()."""
        opts = [o for o in (self.opt_c, self.opt_gf) if o is not None]
        if not opts:
            return
        if self._fused_adam is None:
            self._fused_adam = False
            if all(type(o) is torch.optim.Adam for o in opts):
                try:
                    # the first layers' padded fp32 copies and bf16x3 images are kept current BY the update (ops.FusedAdam
                    # mirrors): the launches that refreshed them at the top of every step are gone (A/B: GRAPES_ADAM_MIRRORS=0)
                    mirrors = {}
                    if _sw("GRAPES_ADAM_MIRRORS", "1") != "0":
                        owned = {id(p) for o in opts for gp in o.param_groups for p in gp["params"]}
                        for fl in self._fl.values():
                            w = fl.conv.lin.weight
                            if id(w) in owned and (fl.split or fl.padded) and w.is_contiguous():
                                mirrors[w] = (fl.W if fl.padded else None, fl.image if fl.split else None)
                    self._fused_adam = ops.FusedAdam(opts, mirrors=mirrors)
                    self._mirrored = {id(w) for w in mirrors}
                except ValueError:
                    self._fused_adam = False
        pend, self._pending_slabs = getattr(self, "_pending_slabs", None), None
        if self._fused_adam:
            self._fused_adam.step(slabs=pend)            # (the classifier's deferred slab sums happen inside the update launch)
            self._mirrors_current = True                 # (from now on the mirrored copies follow the weights by themselves)
        else:
            if pend is not None:
                pend.flush()
            for o in opts:
                o.step()

    def _expand(self, rows, d_m, mark=False, prev_buf=None, remark=None, count=None, stage=None, hop_count=None, finish=None,
                ext=None, ext_out=None):
        """get_neighborhoods of `rows`; in the one-launch form also the next hop's bitmap marks (into prev_buf / g.bits,
        both clean at that point of the step) and the slice re-mark of the current hop (`remark`)."""
        g = self.g
        self._marked = False
        if self.part_adj:
            return g.expand(rows, self.e_cap, d_m=d_m, cap=rows.numel(), want_eoff=True)
        if rows.numel() <= 2048:
            self._marked = mark
            return ops.frontier_expand_fused(self._rp, self._cl, rows, self.e_cap, d_m=d_m, status=g.status,
                                             mark_prev_bits=prev_buf if mark else None, mark_bits=g.bits if mark else None,
                                             num_nodes=g.num_nodes, remark=remark,
                                             count_mult=count[0] if count else None, count_bsum=count[1] if count else None,
                                             slice_stage=stage, count=hop_count, finish=finish, node_ext=ext, node_ext_out=ext_out)
        assert remark is None and count is None and finish is None and ext is None
        eoff, d_e = ops.frontier_offsets(self._rp, rows, d_m=d_m)
        src, dst, _ = ops.frontier_expand(self._rp, self._cl, rows, eoff, self.e_cap, d_m=d_m, status=g.status)
        return src, dst, d_e, eoff

    def _hop_modes(self):
        """-> (fused, staged, counted): which forms of the hop launches this step uses."""
        g, N, B, K, hops, rnd = self.g, self.g.num_nodes, self.B, self.K, self.hops, self.random_sampling
        n_cap = self.n_cap
        fused = (not self.part_adj) and B + K <= 2048
        # slice_adjacency without a launch of its own: the expansion stages the surviving edges, the classifier's graph build
        # (one workgroup per layer graph) assembles the lists
        staged = (fused and self.nall_cap <= 2048 and hops <= 8 and _sw("GRAPES_SLICE_STAGED", "1") != "0")
        # the hop graph's degree counting rides in the expansion (per-edge in-degree atomics whose return value is the entry's
        # slot in its row) and in the compaction (row starts, dinv, segments): the build itself is two launches, not four
        # (measured: products -33 us/step, arxiv -3, Reddit +-0; Cora lost 20 us while its one-workgroup bitmap went through the
        # look-back compaction and gains 12 — 0.705 -> 0.693 ms/step — with the small-bitmap form: GRAPES_HOP_COUNTED_MIN=65536
        # restores the old lower bound on the node count)
        # (... and not above 16.7M nodes: papers100M, 111M nodes, measured 0.647 ms/step counted against 0.617 — the global-id counter
        # tables are 444 MB each there and every per-node read of them misses cache and TLB; GRAPES_HOP_COUNTED_HUGE=1 to A/B)
        counted = (fused and not rnd and n_cap > 2048 and hasattr(g, "hop_counters") and B + K <= 2048 and
                   int(_sw("GRAPES_HOP_COUNTED_MIN", "0")) <= N <= (255 * 65536 if _sw("GRAPES_HOP_COUNTED_HUGE", "0") == "0" else 255 * 65536 * 8) and
                   _sw("GRAPES_HOP_COUNTED", "1") != "0")
        return fused, staged, counted

    # ------------------------------------------------------------------ the step body (captured once)
    def _step_impl(self):
        for _ in self._step_gen():
            pass

    def _step_gen(self):
        """The step as a generator that yields ONCE, at the end of its PRELUDE: the next batch (step_begin), hop 0's expansion,
        compaction and graph build and — for aggregate-first first layers — hop 0's Â[X|ind].  Nothing before the yield reads
        a weight, so the prelude of step t + 1 may run while step t is still training (the prelude pipeline, `_run_pipelined`:
        the two halves are captured as two hipGraphs and replayed on two streams); run straight through it is the whole step."""
        g, N, B, K, hops, num_ind = self.g, self.g.num_nodes, self.B, self.K, self.hops, self.num_ind
        e_cap, n_cap = self.e_cap, self.n_cap
        st = g.status
        targets = self.targets
        ep = self.epoch_t
        ops.set_scratch_lane(self._prelude_lane)          # (the prelude's one-launch scans may run beside another step's)
        if self._loader is not None:      # the step feeds itself: next batch, epoch, target indicators, edge totals (one launch)
            ids, stride, offset = self._loader
            ops.step_begin(ids, self._cursor, stride, offset, targets, ind_code=g.ind_code if num_ind else None,
                           d_epoch=ep if num_ind else None, bit=max(num_ind - 1, 0), counters=self._ctr[:, 2],
                           totals=self.edge_totals)
        elif num_ind:                                                                      # main.py:167-168 (new epoch)
            ops.indicator_mark(g.ind_code, targets, 0, num_ind - 1, d_epoch=ep, advance_epoch=True)
        rnd = self.random_sampling
        ev = self.evaluate
        if not rnd:
            st_gf = self._fl[id(self.gcn_gf.gcn_layers[0])]
            st_z = None if ev else self._fl[id(self.gcn_z.gcn_layers[0])]
        previous, d_m = targets, None                                                      # main.py:163
        # one-launch expansions carry the bitmap marks and the slice re-marks; they alternate two previous-node bitmaps so
        # that a launch can set the next hop's previous set while it clears this hop's
        fused, staged, counted = self._hop_modes()
        if ev:
            staged = False       # (the evaluation's slices come from each hop's OWN expansion, filtered after its draw: below)
        # the draw's last launch without a tail: its log-prob sum and histogram reset ride in the expansion that follows it
        # (one-launch expansions only; A/B: GRAPES_DRAW_DEFER=0)
        defer_draw = fused and self.B + self.K * self.hops <= 2048 and _sw("GRAPES_DRAW_DEFER", "1") != "0"
        # RCCL request / reply form: the hops' exchanged rows are kept side by side and the classifier's features (main.py:256) are
        # found among them — all_nodes are targets + kept nodes, batch rows of the hops' fetches — instead of being requested
        # again: two collectives per step less (A/B: GRAPES_HALO_REUSE=0)
        halo_reuse = (self.partitioned and self._halo_in_place and not rnd and not self.embed and
                      getattr(g, "can_reuse_rows", False) and _sw("GRAPES_HALO_REUSE", "1") != "0")
        kept_halo = None
        pbuf = [g.prev_bits, g.prev_bits_b] if fused else [g.prev_bits, g.prev_bits]
        hc = g.hop_counters() if counted else None
        hbs = [ops.HopBuild(n_cap, e_cap, targets.device, counters=self._ctr[h]) for h in range(hops)] if counted else None
        # the query lists' ROW EXTENTS travel with their ids (round 5): hop 0's expansion leaves the targets' (rowptr[id], rowptr[id + 1]),
        # every draw writes its kept nodes' next to the ids of  cat(targets, kept)  — the expansions of hops >= 1 then read ids and
        # extents in one round trip instead of two dependent ones (A/B: GRAPES_EXPAND_EXT=0)
        with_ext = fused and defer_draw and previous.numel() <= 2048 and _sw("GRAPES_EXPAND_EXT", "1") != "0"
        tgt_ext = torch.empty(2 * B, dtype=torch.int64, device=targets.device) if with_ext else None
        src, dst, d_e, eoff = self._expand(previous, d_m, mark=True, prev_buf=pbuf[0],     # main.py:180 (hop 0) + its marks
                                           hop_count=(hc, hbs[0]) if counted else None, ext_out=tgt_ext)
        hop_state: List[Dict] = []
        hop_stats = torch.empty((hops, 6), dtype=torch.float32, device=targets.device)     # one statistics row per hop
        kept_list, slices, neigh_list, nbl_list, dnn_list, dnb_list = [], [], [], [], [], []
        stages = []
        # device counters of every graph build in one table: column 2 = edges one aggregation over that graph sums
        ctr = self._ctr
        agg_w = [0] * (2 * hops)                                                           # aggregations per graph
        agg_x = [0] * (2 * hops)                          # ... of which run as aggregation launches (see `reuse` below)
        if not rnd:
            gf1, gf2 = self.gcn_gf.gcn_layers
            z1, z2 = (None, None) if ev else self.gcn_z.gcn_layers
        zstate = None
        for hop in range(hops):                                                            # main.py:178
            cur_prev = pbuf[hop % 2]
            if not self._marked:      # (the fused expansion of the previous iteration has done this hop's marks)
                ops.bitmap_mark_hop(cur_prev, g.bits, g.bits1, previous, eoff, dst, N, d_m=d_m, d_e=d_e, status=st)
            # (the compaction also clears the scratch of the graph build that follows it: one launch less per hop)
            # ... and, with the one-launch expansions, applies the slice marks of this hop (un-mark older samples, mark the
            # newer ones: main.py:241-243 keeps the columns `previous` = targets + the samples of the hop before; the targets
            # stay marked for the whole step, the last marks go when all_nodes is built) and zeroes the survivor counters the
            # hop's expansion fills for slice_filter
            pscr = ops.PreparedGraph.scratch(n_cap, src.numel(), targets.device) if (n_cap > 2048 and not rnd and not counted) else None
            bsum = torch.empty(max(int(ops.lib().grapes_slice_filter_workspace_bytes(e_cap)) // 4, 1), dtype=torch.int32,
                               device=targets.device) if (fused and not staged and not ev) else None
            sstage = ops.slice_stage(e_cap, targets.device) if staged else None
            rm_lists = dict(mult=g.mult, unmark=kept_list[hop - 2] if hop >= 2 else None,
                            mark=(targets, None) if hop == 0 else kept_list[hop - 1])
            batch, neigh, nbl, counts, cand_pos = ops.frontier_compact(
                g.bits, g.bits1, cur_prev, N, n_cap, node_map=g.node_map, status=st,
                ind_code=g.ind_code if num_ind else None, d_epoch=ep, ind_bit=hop, want_cand_pos=True,
                zero=(list(pscr[2]) if pscr is not None else []) + ([(hbs[hop].csr_dst, e_cap)] if counted else []) +
                     ([(bsum, bsum.numel())] if bsum is not None else []),
                remark=rm_lists if (fused and not ev) else None,                           # main.py:183-194 (+ 191)
                degrees=(hc, hbs[hop]) if counted else None)
            d_nb, d_nn = counts[0:1], counts[1:2]
            neigh_list.append(neigh); nbl_list.append(nbl); dnn_list.append(d_nn); dnb_list.append(d_nb)
            hid = batch
            if (not rnd) and self.partitioned:
                hid = None
                if self._halo_in_place:            # where this hop's rows will sit in the exchanged buffer (+ their indicator words)
                    hid, self._halo_code, _ = g.halo_positions(batch, d_nb, batch.numel(), ind_code=g.ind_code if num_ind else None,
                                                               tag="h%d" % hop)
            prep = ax_pre = None
            if not rnd:
                pf_rows = (self._prefetch_X, self.F) if (self._prefetch_X is not None and hid is batch and st_gf.agg_first) else None
                # transform-first layers aggregate [n, hidden] activations, not rows of X: their graph's head records are taken over
                # LOCAL ids (head_ids = 0, 1, 2, ...) and drive the record form of that aggregation (one dependent trip per row)
                local_heads = (hid is batch) and not st_gf.agg_first
                if local_heads:
                    if getattr(self, "_iota", None) is None:
                        self._iota = torch.arange(n_cap, dtype=torch.int32, device=targets.device)
                    hid = self._iota
                if counted:
                    prep = ops.PreparedGraph.counted(src, dst, hbs[hop], n_cap, d_nb, d_e, g.node_map, status=st, head_ids=hid,
                                                     prefetch=pf_rows, head_local=local_heads)
                else:
                    prep = ops.PreparedGraph(src, dst, n_cap, d_n=d_nb, d_e=d_e, status=st, src_grouped=True,
                                             items_fwd=False, node_map=g.node_map,                 # main.py:195 relabel inside
                                             head_ids=hid, counters=ctr[hop], scratch=pscr, head_local=local_heads,
                                             # the rows of X the fused gather-SpMM reads ~20 us later, fetched into the Infinity
                                             # Cache by spare workgroups of the build's first launch
                                             prefetch=pf_rows)
                if hop == 0 and st_gf.agg_first and not self.partitioned:
                    # main.py:199-204 + the aggregation of the sampler net's first layer: Â [X | ind | 0] of hop 0 depends on the
                    # batch only — the last launch of the prelude
                    ax_pre = ops.gcn_aggregate_gather(self.Xp, batch, prep, g.ind_code if num_ind else None, 0, num_ind,
                                                      d_epoch=ep if num_ind else None, F=self.F)
            if hop == 0:
                # ---- end of the PRELUDE: nothing above reads a weight (see _step_gen's docstring)
                ops.set_scratch_lane(0)
                yield
                fls = list(self._fl.values())              # weight images of the first layers (strided copies; no-ops when
                if self._mirrors_current:                  # (kept current by the optimiser launch: _optim_step)
                    fls = [fl for fl in fls if id(fl.conv.lin.weight) not in self._mirrored]
                sp = [fl for fl in fls if fl.split]        # F + num_ind is a multiple of 4)
                if 2 <= len(sp) <= 4 and _sw("GRAPES_IMAGES_ONE_LAUNCH", "1") != "0":
                    ops.weight_split_images([fl.conv.lin.weight.detach() for fl in sp], [fl.image for fl in sp],
                                        [fl.W if fl.padded else None for fl in sp])
                    fls = [fl for fl in fls if not fl.split]
                for fl in fls:
                    fl.refresh()
                if self.embed:
                    ops.fill(self.X.grad.view(-1), 0.0)                  # optimizer_c.zero_grad() (main.py:263) for the embeddings
            if rnd:
                # main.py:206-220 with constant logits: the hop graph is never built (no net reads it), the draw is uniform
                res = ops.gumbel_topk(self._rnd_logits, K, logit_index=nbl, candidate_ids=neigh, n=n_cap, d_n=d_nn,
                                      philox_seed=self.seed, d_philox_offset=self.philox_off, want_stats=True,
                                      prefix_ids=targets, stats_out=hop_stats[hop], defer_finish=defer_draw,
                                      ext=(self._rp, tgt_ext) if with_ext else None)
                kept_list.append((res["kept_ids"], res["kept_count"]))                     # main.py:221
            if not rnd:
                fuse_keys = _sw("GRAPES_FUSED_KEYS", "0") != "0"     # measured: 23.7 + 25.2 us vs 4.9 + 14.8 + 23.4 unfused — off
                # main.py:227: at hop 0 the log-Z net sees data.x[batch_nodes] — the rows the sampler net aggregates, minus the
                # indicator columns — so its  Â X  is the leading F columns of the sampler net's  Â [X | ind]: read in place (row
                # stride F + ind) instead of a second gather-SpMM over the same rows (columns F .. ceil4(F) of that view hold
                # aggregated indicator values; the log-Z weight image is zero there)
                reuse = ((not ev) and hop == 0 and (not self.partitioned or self._halo_in_place) and st_gf.agg_first and st_z.agg_first and
                         _sw("GRAPES_FUSED_HEAD", "1") != "0" and
                         ops.split_gemm_available(n_cap, st_z.Kp, z1.lin.weight.shape[0]) and
                         # (f_in > 112 — arxiv, papers100M — has the gate-word backward only: the strided view needs it)
                         (st_z.Kp <= 112 or _sw("GRAPES_GATE_BITS", "1") != "0"))
                # ... and the two nets' 1-wide heads are then aggregated over the hop graph by ONE launch
                pair_heads = reuse and not fuse_keys and _sw("GRAPES_HEAD_PAIR", "1") != "0"
                gemm_pair = pair_heads and _sw("GRAPES_GEMM_PAIR", "1") != "0"
                # transform-first nets (Reddit, Cora's wide frontiers): at hop 0 the sampler net's and the log-Z net's X Wᵀ read the
                # same rows — one launch over both, its last partial round cut along K (A/B: GRAPES_TSPLIT_FWD_DUAL=0)
                h_gf = h_z = None
                if ((not ev) and hop == 0 and not st_gf.agg_first and not st_z.agg_first and st_gf.image is not None and st_z.image is not None and
                        gf1.out_channels == z1.out_channels and gf1.out_channels % 4 == 0 and batch.numel() >= int(_sw("GRAPES_TSPLIT_FWD_DUAL_MIN", "8192")) and
                        not self.partitioned and self.peers is None and _sw("GRAPES_TSPLIT_FWD_DUAL", "1") != "0" and
                        (self.F + num_ind + 31) // 32 == (self.F + 31) // 32):      # (the nets share the K steps)
                    h_gf, h_z = ops.linear_fwd_gathered_tail(self.Xp, self.F, batch, [st_gf.image, st_z.image], gf1.out_channels,
                                                             [self.g.ind_code if num_ind else None, None], [num_ind, 0],
                                                             d_epoch=ep if num_ind else None, d_n=prep.d_n)
                ff = self._first_fwd(gf1, batch, prep, num_ind, ep, head=gf2, defer_head=fuse_keys or pair_heads,   # main.py:199-210
                                     pair=(st_z, z1, z2) if gemm_pair else None, ax=ax_pre, h_pre=h_gf,
                                     keep=(hop, hops) if halo_reuse else None)
                if halo_reuse:
                    kept_halo = self._halo
                    if "base" not in kept_halo:
                        halo_reuse, kept_halo = False, None        # (a first layer that did not fetch through fetch_halo)
                    elif hop == 0:                                 # the targets are batch rows of hop 0 (node_map: its relabel)
                        g.note_rows(kept_halo, targets, None, hid, node_map=g.node_map, batch=batch, d_n_batch=d_nb)
                x, act1, logit = ff[:3]
                agg_w[hop] += 2
                agg_x[hop] += 2
                z_pre = None
                if pair_heads and isinstance(logit, tuple):
                    xz = x[:, :st_z.Kp]
                    if len(ff) == 4:                   # both nets' first layers came out of ONE launch
                        xz, zact, zhw = ff[3]
                    elif self._gate_bits(xz, st_z, z1):
                        zact, zhw = ops.linear_relu_head_fwd_bits(xz, st_z.weight, z1.bias, z2.lin.weight, d_n=prep.d_n)
                    else:
                        zact, zhw = ops.linear_bias_act_head_fwd_strided(xz, st_z.weight, z1.bias, True, z2.lin.weight, d_n=prep.d_n)
                    logit, zout = ops.gcn_aggregate_narrow_pair(logit[1], zhw, prep, gf2.bias, z2.bias)    # Â (act w2ᵀ) + b2, twice
                    z_pre = (xz, zact, zout)
                # exact-k draw over the neighbour candidates (main.py:213-220); logits are read through nb_local
                agg = None
                if isinstance(logit, tuple):       # logits = Â (act w2ᵀ) + b2 formed by the draw's first launch, with the keys
                    agg, logit = (logit[1].view(-1), prep, gf2.bias, cand_pos), None
                res = ops.gumbel_topk(None if agg is not None else logit.view(-1), K, logit_index=nbl, candidate_ids=neigh, n=n_cap,
                                      d_n=d_nn, philox_seed=self.seed, d_philox_offset=self.philox_off, want_stats=True,
                                      prefix_ids=targets, stats_out=hop_stats[hop], agg=agg, defer_finish=defer_draw and agg is None,
                                      mode=1 if ev else 0,                                     # (eval.py:126-130: greedy)
                                      ext=(self._rp, tgt_ext) if (with_ext and agg is None) else None)
                if agg is not None:
                    logit = res["logits"]                                                      # [n_cap, 1]
                kept_list.append((res["kept_ids"], res["kept_count"]))                         # main.py:221
                if halo_reuse:                  # the kept nodes are candidates, i.e. batch rows, of this hop's fetch
                    g.note_rows(kept_halo, res["kept_ids"], res["kept_count"], hid, idx_a=res["kept_pos"], idx_b=nbl)
                if hop == 0 and not ev:                                                        # main.py:223-228
                    if z_pre is not None:
                        xz, zact, zout = z_pre
                    elif reuse:
                        xz = x[:, :st_z.Kp]
                        if self._gate_bits(xz, st_z, z1):
                            zact, zhw = ops.linear_relu_head_fwd_bits(xz, st_z.weight, z1.bias, z2.lin.weight, d_n=prep.d_n)
                        else:
                            zact, zhw = ops.linear_bias_act_head_fwd_strided(xz, st_z.weight, z1.bias, True, z2.lin.weight,
                                                                             d_n=prep.d_n)
                        zout = ops.gcn_aggregate_fwd(zhw, prep, z2.bias, False)
                    else:
                        xz, zact, zout = self._first_fwd(z1, batch, prep, 0, ep, halo=self._halo if self.partitioned else None,
                                                         head=z2, h_pre=h_z)              # zout's mean: in step_losses
                    zstate = dict(x=xz, act=zact, prep=prep, d_nb=d_nb, batch=batch, zout=zout)
                    agg_w[hop] += 2
                    agg_x[hop] += 1 if reuse else 2
                hop_state.append(dict(x=x, act1=act1, logit=logit, prep=prep, nbl=nbl, mask=res["mask"], d_nn=d_nn, cand_pos=cand_pos,
                                      stats=res["stats"]))
            batch_next, d_m_next = res["union_ids"], res["union_count"]                    # main.py:236-238
            # main.py:241-243: the columns kept are `previous` = targets + the samples of the hop before.  The targets stay
            # marked for the whole step; the older samples are un-marked and the newer ones marked in one launch (they are
            # disjoint); the last marks go when all_nodes is built below.  The hop's prev_bits are done with, too.
            if hop + 1 == hops and getattr(self, "_rider_hook", None) is not None:
                # from here on the step's launches are the classifier's: a handful of workgroups each on an idle chip — the NEXT
                # step's recorded prelude rides in them (_capture_pipeline)
                self._rider_hook()
            if ev:
                # eval.py:140-142: slice_adjacency(rows = previous_nodes, cols = the new layer) — this hop's OWN expansion (still
                # in src / dst), filtered by membership in batch_next: mark, ordered filter, un-mark (no host read)
                ops.slice_mark(g.mult, batch_next, d_c=d_m_next)
                ev_slice = ops.slice_filter(g.mult, src, dst, min(e_cap, (B + K) * (B + K)), d_e=d_e, status=st)
                if not fused:
                    ops.slice_mark(g.mult, batch_next, unmark=True, d_c=d_m_next)
            if fused:                 # the expansion of the next previous_nodes also clears this hop's previous-set bitmap
                # and counts the slice survivors of its own edges against the marks made at the top of the hop
                # (evaluation: the un-mark of the slice just taken rides here instead — this launch reads no mark)
                src, dst, d_e, eoff = self._expand(batch_next, d_m_next, mark=hop + 1 < hops, prev_buf=pbuf[(hop + 1) % 2],
                                                   remark=dict(mult=g.mult if ev else None, unmark=(batch_next, d_m_next) if ev else None,
                                                               clear=(previous, d_m), clear_bits=cur_prev),
                                                   count=None if ev else (g.mult, bsum), stage=sstage,     # (the last one only feeds the slice)
                                                   hop_count=(hc, hbs[hop + 1]) if (counted and hop + 1 < hops) else None,
                                                   finish=res.get("finish"),               # (+ the end of this hop's draw)
                                                   ext=res.get("union_ext"))
            else:
                ops.slice_remark(g.mult, unmark=rm_lists["unmark"], mark=rm_lists["mark"], clear=(previous, d_m),
                                 clear_bits=cur_prev)
                src, dst, d_e, eoff = self._expand(batch_next, d_m_next, mark=hop + 1 < hops, prev_buf=cur_prev)
            if staged:                # outputs of the classifier's graph build below (which assembles them from the stage)
                kcap = min(e_cap, (B + K) * (B + K))
                ksrc = torch.empty(kcap, dtype=torch.int32, device=targets.device)
                kdst = torch.empty(kcap, dtype=torch.int32, device=targets.device)
                kcnt = torch.empty(1, dtype=torch.int32, device=targets.device)
                stages.append((sstage, d_e, e_cap))
            elif ev:
                ksrc, kdst, kcnt = ev_slice
            else:
                ksrc, kdst, kcnt = ops.slice_filter(g.mult, src, dst, min(e_cap, (B + K) * (B + K)), d_e=d_e, status=st,
                                                    bsum=bsum if fused else None)
            slices.append((ksrc, kdst, kcnt))
            previous, d_m = batch_next, d_m_next                                           # main.py:247
        # ---- final relabel + classifier (main.py:252-261)
        marks = [(targets, None)] + [(kept, cnt) for kept, cnt in kept_list]                # main.py:221,252
        if len(marks) <= 4 and sum(m_[0].numel() for m_ in marks) <= 4096:     # a thousand ids: sort them in one workgroup
            alln, counts = ops.union_sorted(marks, N, self.nall_cap, node_map=g.node_map, status=st, unmark_mult=None if ev else g.mult)
        else:
            for i in range(0, len(marks), 4):
                ops.bitmap_mark_lists(g.bits, g.bits1, marks[i:i + 4], N, status=st, unmark_mult=g.mult)
            alln, _, _, counts = ops.frontier_compact(g.bits, g.bits1, None, N, self.nall_cap, node_map=g.node_map,
                                                      status=st)
        d_na = counts[0:1]
        hid = None if self.partitioned else alln
        if self.nall_cap <= 2048 and len(slices) <= 8:      # the per-layer subgraphs of the classifier in ONE launch
            preps = ops.PreparedGraph.small_batch(slices, self.nall_cap, d_n=d_na, status=st, node_map=g.node_map,   # main.py:254
                                                  head_ids=hid, counters=ctr[hops:], stages=stages if staged else None)
        else:
            preps = [ops.PreparedGraph(ksrc, kdst, self.nall_cap, d_n=d_na, d_e=kcnt, status=st, src_grouped=True,
                                       node_map=g.node_map, head_ids=hid, counters=ctr[hops + i])
                     for i, (ksrc, kdst, kcnt) in enumerate(slices)]
        layers = list(self.gcn_c.gcn_layers)
        used = [preps[-i] for i in range(1, len(layers))] + [preps[0]]                     # gcn.py:31,35
        first_relu = len(layers) > 1
        first_fused = (not self.partitioned) or (len(layers) > 1 and self.F < layers[0].out_channels)
        # modules/gcn.py:33,37: F.dropout on every layer's output, the logits included, from the sampler's Philox stream (the
        # eager step's module draws the same masks: modules/gcn.py GCN.philox_dropout)
        pdrop = self.dropout
        keeps: List[Optional[torch.Tensor]] = []

        def drop(t):
            if not pdrop:
                keeps.append(None)
                return t
            y, keep = ops.dropout_fwd(t, pdrop, philox_seed=self.seed, d_philox_offset=self.philox_off, d_n=d_na)
            keeps.append(keep)
            return y
        x_rows = self.g.rows_from_kept(kept_halo, alln, d_na) if (halo_reuse and kept_halo is not None) else None
        if first_fused:
            xc, a1 = self._first_fwd(layers[0], alln, used[0], 0, ep, relu=first_relu, x_rows=x_rows)     # main.py:256-257
            acts = [xc, drop(a1)]
        elif x_rows is not None:
            acts = [x_rows]
        else:
            acts = [self.g.assemble(self.g.fetch_halo(alln, d_n=d_na))]
        for li in range(len(acts) - 1, len(layers)):
            acts.append(drop(self._conv_fwd(layers[li], acts[-1], used[li], li < len(layers) - 1)))
        for p in used:
            agg_w[hops + next(i for i, q in enumerate(preps) if q is p)] += 1
            agg_x[hops + next(i for i, q in enumerate(preps) if q is p)] += 1
        logits = acts[-1]
        if ev:
            # eval.py:154-155: predictions = argmax(logits)[node_map.map(target_nodes)] (multi-label: the logit > 0 rows, eval.py:58)
            # (one launch: map, row copy, argmax — rows beyond the live count of `logits` are never addressed by a target)
            pred, rows = ops.eval_predict(logits, g.node_map, targets, status=st)
            self.out = dict(pred=pred if self.y.dim() == 1 else (rows > 0), target_logits=rows, n_all=d_na,
                            all_nodes=alln, logits=logits, kept=[k for k, _ in kept_list], kept_counts=[c for _, c in kept_list],
                            sizes=dnn_list, agg_counts=ctr[:, 2], agg_weights=tuple(agg_w), agg_executed=tuple(agg_x),
                            hop_logits=[hs["logit"] for hs in hop_state], nb_local=nbl_list, neighbor_nodes=neigh_list)
            return
        # ---- both losses in one launch: main.py:259-260 (+ the gradient loss_c.backward() starts from), the mean of the
        # log-Z head (main.py:228) and the GFlowNet loss (main.py:272-282)
        # main.py:260-261: + reg_param * sum of the rows' logit variances — part of loss_c and of the GFlowNet cost (main.py:274)
        reg_term = ops.logit_var_reg(logits, self.reg_param, d_n=d_na) if self.reg_param else None
        loss_c, dl, out4 = ops.step_losses(logits, g.node_map, targets, self.y, hop_stats, self.loss_coef,
                                           z_out=None if rnd else zstate["zout"].view(-1),
                                           d_nz=None if rnd else zstate["d_nb"],
                                           log_z_init=self.log_z_init, reinforce=self.reinforce, loss_extra=reg_term)
        if self.reg_param:
            ops.logit_var_reg(logits, self.reg_param, d_n=d_na, dlogits=dl)
        loss_gfn, s, log_z, tot = out4[0], out4[1:2], out4[2], out4[3]
        def classifier_backward():                                                         # main.py:267
            d = dl
            # the layers' few-row weight gradients leave their slab sums to ONE launch at the end (ops.DeferredSlabs)
            deferred = ops.DeferredSlabs() if _sw("GRAPES_DEFER_SLABS", "1") != "0" else None
            for i in range(len(layers) - 1, -1, -1):
                if keeps[i] is not None:                   # d (layer i's output before dropout)
                    d = ops.dropout_bwd(d, keeps[i], pdrop, d_n=d_na)
                if i == 0 and first_fused:
                    self._first_bwd(layers[0], acts[0], acts[1], d, used[0], False, relu=first_relu, defer=deferred)
                    if self.embed:
                        # d loss_c / d X[all_nodes] = Âᵀ((d ⊙ [act > 0]) W) = (Âᵀ(d ⊙ [act > 0])) W  (main.py:256,267 through
                        # data.x[all_nodes]); all_nodes is duplicate-free and X.grad was zeroed above: plain row stores
                        fl0 = self._fl[id(layers[0])]
                        if not fl0.agg_first:
                            raise NotImplementedError("embedding gradient through a transform-first classifier layer")
                        t, _ = ops.gcn_aggregate_bwd(d, used[0], relu_out=acts[1] if first_relu else None, want_bias=False)
                        dxr = ops.linear_bwd_input(t, fl0.weight, d_n=used[0].d_n)
                        ops.scatter_rows(self.X.grad, alln, dxr, d_n=d_na, F=self.F)
                else:
                    d = self._conv_bwd(layers[i], acts[i], acts[i + 1], d, used[i], i < len(layers) - 1, i > 0, False,
                                       defer=deferred)
            if deferred is not None:
                # one GPU, fused Adam, no padded first-layer gradient to publish: the slab sums ride in the optimiser launch
                if (self.grad_sync is None and self._fused_adam is not False and self.opt_c is not None and
                        not any(fl.publishes for fl in self._fl.values()) and not self.reinforce and
                        _sw("GRAPES_ADAM_SLABS", "1") != "0"):
                    self._pending_slabs = deferred
                else:
                    deferred.flush()
        # A/B (GRAPES_BWD_FORK=1): the classifier's backward chain and the sampler / log-Z nets' backward chain depend on the
        # losses only, not on each other — as two branches of the graph (fork after the loss launch, join before Adam)
        fork = (not rnd) and _sw("GRAPES_BWD_FORK", "0") != "0"
        if fork:
            main_s = torch.cuda.current_stream()
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream()
            self._side.wait_stream(main_s)
            with torch.cuda.stream(self._side):
                classifier_backward()
        multi = False
        z_done = False
        cb_done = False
        # A/B (GRAPES_BWD_CARRY=0): the classifier's backward aggregations as launches of their own, after the sampler nets'
        carry_bwd = (not fork) and (not self.embed) and _sw("GRAPES_BWD_CARRY", "1") != "0"
        if not rnd:
            fi_, fo_ = st_gf.Kp, gf1.out_channels
            multi = hops <= 4 and st_gf.agg_first and fi_ % 4 == 0 and fi_ % 128 != 0 and fo_ % 4 == 0
        # the heads' part of the backward pass for every hop (and the log-Z head) in two launches also when the layers' own
        # weight gradients stay per hop (transform-first layers: Reddit) — instead of zero fill + d logits + aggregation per hop
        heads = multi or ((not rnd) and hops <= 4 and all(hs.get("cand_pos") is not None for hs in hop_state) and
                          _sw("GRAPES_HEAD_BWD_MULTI", "1") != "0")
        dh2s, z_dh2 = None, None
        if heads:
            # the sampler GCN's weights are shared by all hops: per hop only the 1-wide part (d logits, its aggregation),
            # then dW1 / db1 / dW2 of ALL hops from ONE split-K GEMM + ONE slab reduction
            # d log_prob / d logit of every hop (dense, no zero fill) + its by-source aggregation: two launches for all hops
            # (the log-Z head's mean gradient and its aggregation ride along as a fourth graph when there is room)
            z_rides = (not self.reinforce) and hops <= 3
            zs = [zstate] if z_rides else []
            hb = ops.SamplerHeadBwdMulti([hs["logit"].view(-1) for hs in hop_state] + [z["zout"].view(-1) for z in zs],
                                         [hs["mask"] for hs in hop_state] + [None for _ in zs],
                                         [hs["cand_pos"] for hs in hop_state] + [None for _ in zs],
                                         [hs["prep"] for hs in hop_state] + [z["prep"] for z in zs],
                                         d_grad_scale=s, sum_out=gf2.bias.grad, accumulate_sum=False,
                                         mean_sum_out=z2.bias.grad if z_rides else None)
            if carry_bwd:
                # main.py:267 and main.py:287 depend on the losses only: the classifier's few-row backward aggregations ride as
                # extra workgroups of these two launches (ops.carry_backward_aggregations), its GEMMs run between them
                end_riders = getattr(self, "_rider_end_hook", None)
                if end_riders is not None:
                    end_riders()          # (the next step's prelude has been carried by the classifier's forward: detach it)
                ops.carry_backward_aggregations([lambda: hb.launch(1), lambda: hb.launch(2)])
                try:
                    classifier_backward()
                finally:
                    ops.flush_backward_hosts()
                cb_done = True
            else:
                hb.launch(0)
            dh2_all = hb.dh
            dh2s = [dh2_all[h][:hs["logit"].numel()] for h, hs in enumerate(hop_state)]
            z_dh2 = dh2_all[hops][:zstate["zout"].numel()].view(-1, 1) if z_rides else None
        if multi:
            if all(isinstance(hs["act1"], ops.GateBits) for hs in hop_state):
                if (z_dh2 is not None and isinstance(zstate["act"], ops.GateBits) and hops <= 3 and
                        zstate["x"].shape[1] <= st_gf.Kp and _sw("GRAPES_DW_PAIR", "1") != "0"):
                    # ... with the log-Z net's first layer (its own weights, the hop-0 rows) as a second problem of the same launch
                    ops.linear_bwd_weight_bits_pair(
                        [hs["act1"] for hs in hop_state], [hs["x"] for hs in hop_state], dh2s, [hs["prep"].d_n for hs in hop_state],
                        gf2.lin.weight.view(-1), st_gf.weight, gf1.bias, st_gf.grad_direct, gf1.bias.grad, gf2.lin.weight.grad.view(-1),
                        zstate["act"], zstate["x"], z_dh2.view(-1), zstate["prep"].d_n, z2.lin.weight.view(-1), st_z.weight, z1.bias,
                        st_z.grad, z1.bias.grad, z2.lin.weight.grad.view(-1), accumulate=False)
                    z_done = True
                else:
                    ops.linear_bwd_weight_bits_multi([hs["act1"] for hs in hop_state], [hs["x"] for hs in hop_state], dh2s,
                                                     [hs["prep"].d_n for hs in hop_state], gf2.lin.weight.view(-1), st_gf.weight,
                                                     gf1.bias, st_gf.grad_direct, dbias=gf1.bias.grad,
                                                     dw_head=gf2.lin.weight.grad.view(-1), accumulate=False)
            else:
                ops.linear_bwd_weight_gated_multi([hs["act1"] for hs in hop_state], [hs["x"] for hs in hop_state], dh2s,
                                                  [hs["prep"].d_n for hs in hop_state], gf2.lin.weight.view(-1),
                                                  st_gf.grad, dbias=gf1.bias.grad,
                                                  dw_head=gf2.lin.weight.grad.view(-1), accumulate=False)
        # transform-first nets (Reddit, Cora): the hops' (and the log-Z net's) rank-1 backward aggregations — five launches each,
        # independent of each other — as ONE chain of five launches (A/B: GRAPES_R1_MULTI=0)
        dh_r1s, z_r1 = [None] * len(hop_state), None
        if (dh2s is not None and not multi and _sw("GRAPES_R1_MULTI", "1") != "0" and
                all(self._r1_ok(gf1, hs["act1"]) for hs in hop_state)):
            z_in = (z_dh2 is not None and not (self.reinforce or rnd or z_done) and self._r1_ok(z1, zstate["act"]) and
                    zstate["act"].shape[1] == hop_state[0]["act1"].shape[1])
            probs = [dict(act=hs["act1"], dh2=dh2s[h].view(-1), w2=gf2.lin.weight.view(-1), prep=hs["prep"],
                          gate_bits=hs["act1"]._gate_bits, dw_head=gf2.lin.weight.grad.view(-1), dbias=gf1.bias.grad, accumulate=h > 0)
                     for h, hs in enumerate(hop_state)]
            if z_in:
                probs.append(dict(act=zstate["act"], dh2=z_dh2.view(-1), w2=z2.lin.weight.view(-1), prep=zstate["prep"],
                                  gate_bits=zstate["act"]._gate_bits, dw_head=z2.lin.weight.grad.view(-1), dbias=z1.bias.grad,
                                  accumulate=False))
            if 2 <= len(probs) <= 3:
                outs = ops.gcn_aggregate_bwd_rank1_multi(probs)
                dh_r1s = outs[:len(hop_state)]
                z_r1 = outs[len(hop_state)] if z_in else None
        # ... and the weight-gradient GEMMs that follow them (the sampler net's per hop, the log-Z net's) as ONE launch + one slab
        # sum: the slab budget dealt out by the live rows instead of 768 workgroups' worth per launch (A/B: GRAPES_DW_MULTI=0)
        dw_batch = [] if (all(d is not None for d in dh_r1s) and dh_r1s and self._fl[id(gf1)].split and
                          _sw("GRAPES_DW_MULTI", "1") != "0") else None
        for h, hs in enumerate(hop_state if not multi else []):
            acc = h > 0                           # hop 0 writes the .grad buffers; later hops accumulate
            if dh2s is not None:
                self._head_bwd(gf1, gf2, hs["x"], hs["act1"], None, hs["prep"], acc, db2_done=True, dh2=dh2s[h].view(-1, 1),
                               num_ind=num_ind, ep=ep, hop=h, dh_r1=dh_r1s[h], dw_batch=dw_batch)
                continue
            dlog = torch.zeros_like(hs["logit"])
            ops.bernoulli_logprob_bwd(hs["logit"].view(-1), hs["mask"], d_grad_scale=s, logit_index=hs["nbl"],
                                      out=dlog.view(-1), d_n=hs["d_nn"], accumulate_sum=acc,
                                      sum_out=gf2.bias.grad)                                # db2 = sum(dlog)
            self._head_bwd(gf1, gf2, hs["x"], hs["act1"], dlog, hs["prep"], acc, db2_done=True, num_ind=num_ind, ep=ep, hop=h)
        if self.reinforce or rnd or z_done:
            pass                                          # (reinforce: the log-Z net takes no part, its gradients are zeroed below)
        elif z_dh2 is not None:                 # d mean / d pred_z and its aggregation came with the hops' (above)
            self._head_bwd(z1, z2, zstate["x"], zstate["act"], None, zstate["prep"], False, db2_done=True, dh2=z_dh2, dh_r1=z_r1,
                           dw_batch=dw_batch if (z_r1 is not None and self._fl[id(z1)].split) else None)
        else:
            dz = torch.empty_like(zstate["zout"].reshape(-1, 1))          # (the activations may be kept as gate bits only)
            ops.fill(dz.view(-1), d_n=zstate["d_nb"], d_value=s, scale_by_inv_n=1.0,       # d mean / d pred_z
                     sum_out=z2.bias.grad)
            self._head_bwd(z1, z2, zstate["x"], zstate["act"], dz, zstate["prep"], False, db2_done=True)
        if dw_batch:
            if ops.linear_bwd_weight_gathered_multi_ok(self.F, dw_batch):
                ops.linear_bwd_weight_gathered_multi(self.Xp, self.F, dw_batch, d_epoch=ep if num_ind else None)
            else:                                 # (widths of another tile shape: one launch each, as before)
                for q in dw_batch:
                    ops.linear_bwd_weight_gathered(q["dh"], self.Xp, self.F, q["ids"], q["dw"], q["ind_code"], 0, q["num_ind"],
                                                   d_epoch=q["ep"], d_n=q["d_n"], accumulate=q["accumulate"], split=q["split"],
                                                   ind_mask=q["ind_mask"])
        if fork:
            main_s.wait_stream(self._side)
        elif not cb_done:
            classifier_backward()                                                          # main.py:267
        for fl in self._fl.values():
            fl.publish_grad()
        if self.reinforce and not rnd:
            for p in self.gcn_z.parameters():
                p.grad.zero_()
        if self.grad_sync is not None:   # ONE flat all-reduce for the three models
            self.grad_sync([p for m in self._models for p in m.parameters()])
        self._optim_step()                                                                 # main.py:268,289
        # (random_sampling: main.py:272 skips the GFlowNet loss, batch_loss_gfn stays 0 and log_z stays at its initial 0)
        self.out = dict(loss_c=loss_c.detach(), loss_gfn=None if rnd else loss_gfn.detach().reshape(()),
                        log_z=None if rnd else log_z.reshape(()),
                        tot_log_prob=tot, agg_counts=ctr[:, 2], agg_weights=tuple(agg_w), agg_executed=tuple(agg_x),
                        n_all=d_na, kept=[k for k, _ in kept_list], kept_counts=[c for _, c in kept_list],
                        all_nodes=alln, logits=logits, sizes=dnn_list,
                        batch_counts=dnb_list, classifier_layers=len(layers),
                        hop_logits=[hs["logit"] for hs in hop_state], nb_local=nbl_list,
                        neighbor_nodes=neigh_list, hop_stats=hop_stats)

    # ------------------------------------------------------------------ public
    def attach_loader(self, train_ids: torch.Tensor, stride: int = 1, offset: int = 0, epochs: bool = False):
        """Device-side batch loader (before the first step): step_next() then takes the unshuffled sequential chunks of
        `train_ids` (main.py:126) — chunk number cursor * stride + offset, wrapped — without any host-side launch around the
        replayed graph, and keeps running 64-bit totals of the per-graph edge counters in `edge_totals` (one step behind:
        a step adds the counters of the step before it).
        epochs=True: the chunks are exactly the FULL batches of the reference's DataLoader, epoch after epoch — starts 0, B, 2B, …,
        (len // B - 1) B, then 0 again; the ragged last batch of an epoch (main.py:126 keeps it) is not the loader's: the caller
        runs it through an eager trainer WITH ITS OWN graph scratch (the next step's prelude may already be in flight on this
        trainer's).  Default: the wrapped chunks of bench.py (start = chunk * B mod (len - B))."""
        if self.steps_done:
            raise RuntimeError("attach_loader must precede the first step")
        dev = self.g.device
        ids = train_ids.to(device=dev, dtype=torch.int32).contiguous()
        if ids.numel() < self.B:
            raise ValueError("fewer training ids than one batch")
        if epochs:
            # grapes_step_begin wraps at len - B: a list of (full + 1) B entries makes that full * B, i.e. start = (chunk mod full) B
            full = ids.numel() // self.B
            ids = torch.cat([ids[:full * self.B], torch.zeros(self.B, dtype=torch.int32, device=dev)])
        self._loader = (ids, int(stride), int(offset))
        self._cursor = torch.zeros(1, dtype=torch.int32, device=dev)
        self.edge_totals = torch.zeros(self._ctr.shape[0], dtype=torch.int64, device=dev)
        if self._pipeline_ok and self.hops >= 2 and all(self._hop_modes()[i] for i in (0, 2)):
            # Two SETS of everything a step in flight owns besides the models — the graph's scratch tables (bitmaps, TensorMap,
            # slice marks, indicator table + epoch, hop counters), the target buffer, the build counters — so that the prelude
            # of step t + 1 (set (t + 1) % 2) can run INSIDE step t (set t % 2): see _capture_pipeline.  The CSR, X, the labels,
            # the status word, the loader cursor and the edge totals are shared.  (Only the fused + counted hop launches can be
            # recorded as riders: other configurations keep the one-set step.)
            ga = self.g
            gb = DeviceGraph(ga.rowptr, ga.col, ga.num_nodes)
            gb.status = ga.status
            gb._max_degree = ga._max_degree
            self._sets = [_StepSet(ga, self.targets, self.epoch_t, self._ctr),
                          _StepSet(gb, torch.zeros_like(self.targets), gb.epoch_counter(), torch.zeros_like(self._ctr))]
            # the prelude's one-launch scans get their own copy of the per-device sync scratch: they run beside the main part's
            GraphedTrainer._lanes += 1
            self._prelude_lane = GraphedTrainer._lanes

    def step_next(self) -> Dict[str, torch.Tensor]:
        """One training iteration on the next batch of the attached loader."""
        if self._loader is None:
            raise RuntimeError("step_next needs attach_loader")
        return self._run()

    def run_steps(self, k: int, chain: int = 8) -> Dict[str, torch.Tensor]:
        """Enqueues the next `k` training iterations of the attached loader (main.py:157 `for batch in loader`, k turns of it)
        and returns the last one's device tensors; nothing synchronises.  Once the step is captured, `chain` consecutive steps go
        out as ONE hipGraphLaunch (include/grapes_hip.h: step chains — copies of the captured steps' kernel nodes, same arguments,
        same order): between two launches of a replayed graph the queue idles for ~20 us, which a chain pays once per `chain`
        steps.  The same kernels run in the same order as k calls of step_next(): results are bit-identical.  Steps that do not
        fill a chain (and every step of a trainer whose step holds collectives, or that is not captured yet) are step_next()."""
        if self._loader is None:
            raise RuntimeError("run_steps needs attach_loader")
        k, chain = int(k), int(chain)
        out = self.out
        if chain > 1 and k > 1 and self._boundary_ready():
            return self._run_boundary_chained(k)
        while k > 0:
            n = self._chain_ready(min(chain, k))
            if n:
                out = self._run_chain(n)
                k -= n
            else:
                out = self._run()
                k -= 1
        return out

    # ---- steps with collectives between their graph segments (N > 1: the gradient all-reduce; the RCCL request / reply form): the
    # collectives stay host-issued, but the LAST segment of step t and the FIRST segment of step t + 1 have nothing between them —
    # they go out as one hipGraphLaunch (a two-segment chain), one launch boundary per step less.
    def _boundary_ready(self) -> bool:
        if self._sets is None or self._sets[0].G is None or not self._primed or getattr(self, "_chains_off", False):
            return False
        for st in self._sets:
            it = st.G.items
            if st.G.num_collectives < 1 or not (isinstance(it[0], torch.cuda.CUDAGraph) and isinstance(it[-1], torch.cuda.CUDAGraph)):
                return False
        return all(getattr(st.g, "_epoch_dev_used", 0) + 4 < st.g._HOST0 - 2 for st in self._sets)

    def _boundary_chain(self, p: int) -> "_StepChain":
        """[last segment of a step of set p, first segment of the next step (set 1 - p)] as one executable graph."""
        import ctypes as C
        key = ("b", p)
        ch = self._chains.get(key)
        if ch is None:
            segs = [int(self._sets[p].G.items[-1].raw_cuda_graph()), int(self._sets[1 - p].G.items[0].raw_cuda_graph())]
            arr = (C.c_void_p * 2)(*segs)
            h, nodes = C.c_void_p(), C.c_int32(0)
            ops._lib.check(ops.lib().grapes_graph_chain_create(arr, 2, 1, C.byref(h), C.byref(nodes)), "graph_chain_create")
            ch = self._chains[key] = _StepChain(h, int(nodes.value))
        return ch

    def _run_boundary_chained(self, k: int) -> Dict[str, torch.Tensor]:
        lib = ops.lib()
        for j in range(k):
            t = self.steps_done
            cur, nxt = self._sets[t % 2], self._sets[(t + 1) % 2]
            items = cur.G.items
            if j == 0:
                if not self._boundary_ready():              # (an epoch range about to be refilled: plain steps)
                    for _ in range(k):
                        self._run()
                    return self.out
                nxt.g.reserve_device_epochs(1)              # (step t + 1's prelude rides in this step: counted before it is enqueued)
                items[0].replay()                           # (later steps: launched with the previous step's last segment)
            for it in items[1:-1]:
                if isinstance(it, torch.cuda.CUDAGraph):
                    it.replay()
                else:
                    it()
            # Step t + 1's first segment goes out WITH this step's last one only if step t + 1 can be chained as a whole: it carries
            # the prelude of step t + 2 (set `cur` again), whose epoch must not need a refill — this step still reads cur's table
            # (deciding at the top of the next iteration replayed that first segment a second time through _run()).
            chain_next = j + 1 < k and self._boundary_ready() and not cur.g.would_refill(1)
            bch = None
            if chain_next:
                try:
                    bch = self._boundary_chain(t % 2)
                except ops._lib.GrapesHipError:       # a node the chain builder does not copy: one launch per segment from now on
                    self._chains_off, chain_next = True, False
            if chain_next:
                cur.g.reserve_device_epochs(1)
                ops._lib.check(lib.grapes_graph_chain_launch(bch.handle, ops._stream()), "graph_chain_launch")
            else:
                items[-1].replay()
            self._activate(cur)
            self.steps_done += 1
            if not chain_next and j + 1 < k:                # nothing of step t + 1 has been launched: plain steps from here on
                for _ in range(k - j - 1):
                    self._run()
                return self.out
        return self.out

    def _chain_ready(self, n: int) -> int:
        """Steps (0: none) the next chain launch may cover: the step is captured, primed and free of collectives."""
        if n < 2 or self.partitioned or self.grad_sync is not None or getattr(self, "_chains_off", False):
            return 0
        if self._sets is not None:
            if self._sets[0].G is None or not self._primed:
                return 0
            n -= n % 2                                     # the two sets' graphs alternate: whole pairs
            graphs = [st.g for st in self._sets]
        else:
            if self.graph_obj is None:
                return 0
            graphs = [self.g]
        # (an epoch range about to run out is refilled between two steps by note_device_epochs: not inside a chain)
        if any(getattr(x, "_epoch_dev_used", 0) + n >= x._HOST0 - 2 for x in graphs):
            return 0
        return n

    def _chain(self, t: int, n: int) -> "_StepChain":
        """The chain of n steps whose first is step number t (built once per parity of t and length)."""
        import ctypes as C
        key = (t % 2 if self._sets is not None else 0, n)
        ch = self._chains.get(key)
        if ch is None:
            if self._sets is not None:
                segs = self._sets[t % 2].G.raw_graphs() + self._sets[(t + 1) % 2].G.raw_graphs()
                rep = n // 2
            else:
                segs, rep = self.graph_obj.raw_graphs(), n
            arr = (C.c_void_p * len(segs))(*segs)
            h, nodes = C.c_void_p(), C.c_int32(0)
            ops._lib.check(ops.lib().grapes_graph_chain_create(arr, len(segs), rep, C.byref(h), C.byref(nodes)), "graph_chain_create")
            ch = self._chains[key] = _StepChain(h, int(nodes.value))
        return ch

    def prepare_chains(self, k: int, chain: int = 8) -> int:
        """Builds (without launching anything) the chains run_steps(k, chain) will use from the current step on, so that no
        graph is instantiated inside a timed region.  Returns how many were built; 0 while the step is not captured yet."""
        k, chain, t, built = int(k), int(chain), self.steps_done, 0
        if chain > 1 and k > 1 and self._boundary_ready():     # steps with collectives: the two boundary chains
            built = sum(("b", p) not in self._chains for p in (0, 1))
            self._boundary_chain(0); self._boundary_chain(1)
            return built
        while k > 0:
            n = self._chain_ready(min(chain, k))
            if not n:
                if self._chain_ready(2) == 0:
                    break
                n = 1                                      # (an odd remainder: a single step)
            else:
                built += (t % 2 if self._sets is not None else 0, n) not in self._chains
                self._chain(t, n)
            k -= n
            t += n
        return built

    def _run_chain(self, n: int) -> Dict[str, torch.Tensor]:
        t = self.steps_done
        try:
            ch = self._chain(t, n)
        except ops._lib.GrapesHipError:               # (graph_chain_create met a node type it does not copy: ADVICE r04)
            self._chains_off = True                   # run_steps keeps working, one launch per step
            for _ in range(n):
                self._run()
            return self.out
        if self._sets is not None:                # (counted before the launch; _chain_ready has made sure that no refill is due)
            for st in self._sets:
                st.g.reserve_device_epochs(n // 2)
        ops._lib.check(ops.lib().grapes_graph_chain_launch(ch.handle, ops._stream()), "graph_chain_launch")
        if self._sets is not None:
            self._activate(self._sets[(t + n - 1) % 2])
        else:
            self.g.note_device_epochs(n)
        self.steps_done += n
        return self.out

    def step(self, target_nodes: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Enqueues one training iteration for `target_nodes` (exactly batch_size ids).  Returns device
        tensors; nothing synchronises.  The first calls run eagerly, then the step is captured."""
        if self._loader is not None:
            raise RuntimeError("a loader is attached: use step_next()")
        if target_nodes.numel() != self.B:
            raise ValueError(f"the captured step has a fixed batch size of {self.B}")
        self.targets.copy_(target_nodes.to(device=self.g.device, dtype=torch.int32), non_blocking=True)
        return self._run()

    def _activate(self, st: "_StepSet"):
        self.g, self.targets, self.epoch_t, self._ctr = st.g, st.targets, st.epoch_t, st.ctr
        self.out = st.out

    def _capture_pipeline(self):
        """The prelude pipeline: step t's hipGraph = the MAIN part of step t (set t % 2) with the PRELUDE of step t + 1 (the other
        set: next batch, hop 0's expansion, compaction, graph build and gather-SpMM — nothing in it reads a weight) riding as
        extra workgroups in its hop-1 launches of the same kernels (include/grapes_hip.h: riders).  Each set's prelude is
        RECORDED once — its buffers are ordinary allocations that live as long as the trainer — and attached while the other
        set's main part is captured, from the last expansion on: the classifier's forward launches (final expansion, all_nodes
        union, per-layer graph build, gather-SpMM, aggregation) are a handful of workgroups each on an idle chip, so the riders
        cost next to nothing there (carried by the hop-1 launches of the same kernels they competed for the same resource and
        the pairs took nearly the sum of their parts: profiles/r04_pipeline_ab.txt).  One graph launch per step on one stream: a second stream or a graph branch would tax every
        dispatch of the first (profiles/r04_overlap_probe.txt)."""
        import ctypes as C
        lib = ops.lib()
        torch.cuda.synchronize()
        for st in self._sets:
            self._activate(st)
            st.gen = self._step_gen()
            ops._lib.check(lib.grapes_rider_record_begin(), "rider_record_begin")
            ops.rider_keep(True)
            # every allocation made while recording stays reserved for the trainer's lifetime: the recorded launches are
            # baked into the OTHER set's hipGraph with raw pointers; a block that returned to the caching allocator could be handed to
            # someone else and every replay would write into it.  (The locals' snapshot below keeps the tensors too: belt and braces.)
            if getattr(self, "_rec_pool", None) is None:
                self._rec_pool = torch.cuda.MemPool() if hasattr(torch.cuda, "MemPool") else None
            pool_ctx = torch.cuda.use_mem_pool(self._rec_pool) if self._rec_pool is not None else contextlib.nullcontext()
            try:
                with pool_ctx:
                    next(st.gen)               # allocates the prelude's buffers, launches nothing
            finally:
                st.program = int(lib.grapes_rider_record_end())
                st.scratch = ops.rider_keep(False)      # (the wrappers' workspaces: the recorded launches use them every step)
            if st.program < 0:
                raise ops._lib.GrapesHipError("rider recording failed")
            st.handoff = dict(st.gen.gi_frame.f_locals)
        pool = torch.cuda.graph_pool_handle()
        for st, other in ((self._sets[0], self._sets[1]), (self._sets[1], self._sets[0])):
            self._activate(st)
            st.G = SegmentedGraph()
            st.G._pool = pool
            hooked = [o for o in (self.g, self.grad_sync) if hasattr(o, "run_collective")]
            for o in hooked:
                o.run_collective = st.G.run_collective

            def body(st=st, other=other):
                # early phase: only the other set's step_begin may ride (one more workgroup of this step's first expansion) ...
                ops._lib.check(lib.grapes_rider_attach(other.program, 1, ops._stream()), "rider_attach")
                released = []

                def hook():           # ... the rest from the last expansion on (called by _step_gen in front of it)
                    ops._lib.check(lib.grapes_rider_release(ops._stream()), "rider_release")
                    released.append(True)
                self._rider_hook = hook
                ended = []

                def end():            # whatever nobody has carried by now is issued on its own; the program is detached
                    if ended:
                        return
                    if not released:
                        hook()        # (a step without that point: the prelude runs on its own, here)
                    paired = C.c_int32(0)
                    alone = lib.grapes_rider_detach(ops._stream(), C.byref(paired))
                    ended.append((int(paired.value), int(alone)))
                self._rider_end_hook = end        # (called by _step_gen once the classifier's forward is through, if it needs the riders)
                try:
                    for _ in st.gen:
                        pass
                finally:
                    self._rider_hook = None
                    self._rider_end_hook = None
                    end()
                if ended[0][1] < 0:
                    raise ops._lib.GrapesHipError(f"rider_detach failed ({ended[0][1]})")
                st.riders = ended[0]              # launches of the other set's prelude that rode / ran on their own
            st.G.record(body)
            st.out = self.out
            st.gen = None
        self.graph_obj = self._sets[0].G
        self._primed = False

    def _run_pipelined(self) -> Dict[str, torch.Tensor]:
        t = self.steps_done
        cur, nxt = self._sets[t % 2], self._sets[(t + 1) % 2]
        if not self._primed:                      # the first pipelined step: nobody has carried its prelude
            cur.g.reserve_device_epochs(1)
            ops._lib.check(ops.lib().grapes_rider_launch(cur.program, ops._stream()), "rider_launch")
            self._primed = True
        # step t + 1's prelude (its step_begin advances nxt's device epoch) rides in this graph: its epoch is counted — and a range
        # that has run out refilled — BEFORE the graph is enqueued, stream-ordered behind nxt's previous step (refilling
        # afterwards cleared the indicator table between that step's prelude and its main part)
        nxt.g.reserve_device_epochs(1)
        cur.G.replay()                            # step t's main part + step t + 1's prelude
        self._activate(cur)
        self.steps_done += 1
        return self.out

    def _run(self) -> Dict[str, torch.Tensor]:
        if self._sets is not None:
            if self._sets[0].G is not None:
                return self._run_pipelined()
            if self._want_capture and self.steps_done >= self.eager_steps:
                self._capture_pipeline()
                return self._run_pipelined()
            self._activate(self._sets[self.steps_done % 2])     # eager warm-up: one step on each set
        if self.graph_obj is not None:
            self.graph_obj.replay()
            if self.partitioned:
                self.g.exchanged_bytes += self._bytes_per_step
        elif self._want_capture and self.steps_done >= self.eager_steps:
            self.graph_obj = SegmentedGraph()
            hooked = [o for o in (self.g, self.grad_sync) if hasattr(o, "run_collective")]
            for o in hooked:
                o.run_collective = self.graph_obj.run_collective
            b0 = self.g.exchanged_bytes if self.partitioned else 0
            self.graph_obj.record(self._step_impl)
            self._bytes_per_step = (self.g.exchanged_bytes - b0) if self.partitioned else 0
            self.graph_obj.replay()
        else:
            if self.partitioned and self.auto_calibrate and self.steps_done == 0:
                self.g.calibrating = True
            self._step_impl()
            if self.partitioned and self.auto_calibrate and self.steps_done == 1:
                self.g.calibrate()                       # halo slot size from the warm-up steps (one host read)
        if self._sets is not None:
            self._sets[self.steps_done % 2].out = self.out
        self.steps_done += 1
        self.g.note_device_epochs(1)
        return self.out

    def __del__(self):
        # the recorded prelude programs of the two scratch sets (riders.hip keeps them in a table: they would leak across --runs)
        try:
            for st in (self._sets or []):
                prog = getattr(st, "program", None)
                if prog is not None and prog >= 0:
                    ops.lib().grapes_rider_free(int(prog))
                    st.program = -1
        except Exception:           # (interpreter shutdown: the library may be gone)
            pass

    def check(self):
        """One host read of the device status word: raises if any hop overflowed its capacity."""
        self.g.check_status("captured step (raise e_cap)")

    @staticmethod
    def edges_aggregated(out: Dict) -> int:
        """Edges summed by the step's aggregations: per graph build (agg_counts) times the aggregations over it."""
        c = out["agg_counts"].to("cpu", torch.int64)
        return int((c * torch.tensor(out["agg_weights"], dtype=torch.int64)).sum().item())
