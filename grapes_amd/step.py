"""A8 — one GRAPES training iteration (reference main.py:157-291) with the graph, the features
and every intermediate resident in HBM.

Control flow, ordering contracts and update order follow the reference line by line (cited
inline); what differs is where things live and how many host round-trips there are:
  * the adjacency and X never leave the device (reference: SciPy CSR + CPU tensors, H2D per hop);
  * the O(N) boolean masks of main.py:183-190,252 become a two-level bitmap compaction;
  * the N x (hops+1) indicator matrix of main.py:140,167 becomes an epoch-tagged code per node;
  * slice_adjacency(rows=batch_nodes, cols=previous_nodes) of hop h and get_neighborhoods of hop
    h+1 expand the SAME rows (main.py:236-247 then :180), so one expansion serves both;
  * one device->host read per hop (three sizes + status word) instead of the reference's
    per-hop mask D2H + SciPy round-trips.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .graph import DeviceGraph
from .modules.utils import sample_neighborhoods_from_probs


class GrapesTrainer:
    def __init__(self, graph: DeviceGraph, X: torch.Tensor, y: torch.Tensor, gcn_c: nn.Module, gcn_gf: nn.Module,
                 gcn_z: nn.Module, *, sampling_hops: int = 2, num_samples: int = 16, use_indicators: bool = True,
                 loss_coef: float = 1e4, log_z_init: float = 0.0, reg_param: float = 0.0,
                 random_sampling: bool = False, reinforce_baseline: bool = False,
                 optimizer_c: Optional[torch.optim.Optimizer] = None,
                 optimizer_gf: Optional[torch.optim.Optimizer] = None, e_cap: Optional[int] = None,
                 philox_seed: Optional[int] = None, grad_sync: Optional[Callable] = None):
        if X is None:
            if not hasattr(graph, "features"):
                raise ValueError("X may only be omitted with a dist.PartitionedGraph (which owns its feature shard)")
        elif not X.is_cuda:
            raise ops._lib.GrapesHipError("X must be resident in HBM (cuda tensor)")
        self.g, self.X, self.y = graph, (None if X is None else X.contiguous()), y
        # --embed_nodes (main.py:89-100): X is an nn.Parameter that optimizer_c owns.  Only the CLASSIFIER's input is gathered
        # with a gradient: what the GFlowNet loss would add to X.grad (through the sampler / log-Z nets' inputs) is never used
        # by the reference either — optimizer_gf does not own the embeddings and optimizer_c.zero_grad() clears it before the
        # next classifier backward (main.py:263-289) — so the parameter updates are identical.
        self.embed = isinstance(X, nn.Parameter) and X.requires_grad
        self.F = X.shape[1] if X is not None else graph.feature_dim
        self.gcn_c, self.gcn_gf, self.gcn_z = gcn_c, gcn_gf, gcn_z
        self.hops, self.K = sampling_hops, num_samples
        self.num_ind = sampling_hops + 1 if use_indicators else 0        # main.py:104-107
        if self.num_ind > 8:
            raise ValueError("at most 7 sampling hops with indicators")
        self.loss_coef, self.log_z_init, self.reg_param = loss_coef, log_z_init, reg_param
        self.random_sampling, self.reinforce_baseline = random_sampling, reinforce_baseline
        self.opt_c, self.opt_gf = optimizer_c, optimizer_gf
        self._fused_opt = {}            # id(optimizer) -> ops.FusedAdam | False (_opt_step)
        self.e_cap = e_cap
        self.epoch = 0
        self.philox_seed = philox_seed
        self.philox_offset = 0
        self.grad_sync = grad_sync      # multi-GPU: all-reduce of the gradients before each optimiser step
        if y is not None:
            self.loss_fn = nn.CrossEntropyLoss() if y.dim() == 1 else nn.BCEWithLogitsLoss()   # main.py:120-123
        if philox_seed is not None and gcn_c is not None and getattr(gcn_c, "dropout", 0.0) and hasattr(gcn_c, "philox_dropout"):
            # dropout masks from the same counter stream as the draws (a captured step with this seed then sees the same masks)
            def _counters(n_elements):
                off = self.philox_offset
                self.philox_offset += (int(n_elements) + 3) // 4
                return self.philox_seed, off
            gcn_c.philox_dropout = _counters

    # ------------------------------------------------------------------
    def _caps(self, m: int):
        g = self.g
        if self.e_cap is not None:
            e_cap = self.e_cap
        else:
            # over a partitioned adjacency e_cap sizes the all-to-all reply slots (stride = 2 cap + e_cap), which must be equal
            # on all ranks: derive it from a row count every rank agrees on (ragged last batches differ from rank to rank)
            mm = g._common_cap(m) if (hasattr(g, "_common_cap") and not getattr(g, "adjacency_replicated", False)) else m
            e_cap = max(1 << 16, min(mm * max(g.max_degree, 1), 1 << 22))
        return e_cap, e_cap + m + 1

    def _uniforms(self, hop: int, n: int, uniforms_fn):
        if uniforms_fn is not None:
            return uniforms_fn(hop, n)
        if self.philox_seed is not None:
            u = ops.philox_uniform(n, self.philox_seed, self.philox_offset, self.g.device)
            self.philox_offset += (n + 3) // 4
            return u
        return None   # sampler draws torch.rand(n) on the device (reference behaviour)

    def _expand(self, rows: torch.Tensor, e_cap: int, mark: bool = True):
        g = self.g
        if hasattr(g, "expand"):            # dist.PartitionedGraph: rows come from their owners (all-to-all)
            self._eoff, self._marked = None, False
            return g.expand(rows, e_cap)
        if mark and rows.numel() <= 2048 and e_cap < 0x7fffffff // 256:
            # one launch: row-length scan + expansion + the hop's three bitmap marks (previous set, queried rows that have edges,
            # every neighbour) — the captured step's form; the eager loop used five launches for this
            src, dst, d_e, eoff = ops.frontier_expand_fused(g.rowptr, g.col, rows, e_cap, status=g.status,
                                                            mark_prev_bits=g.prev_bits, mark_bits=g.bits, num_nodes=g.num_nodes)
            self._eoff, self._marked = eoff, True
            return src, dst, d_e
        eoff, d_e = ops.frontier_offsets(g.rowptr, rows)
        src, dst, _ = ops.frontier_expand(g.rowptr, g.col, rows, eoff, e_cap, status=g.status)
        self._eoff, self._marked = eoff, False
        return src, dst, d_e

    def _features(self, ids: torch.Tensor, epoch: int = 0, num_ind: int = 0, differentiable: bool = False) -> torch.Tensor:
        """[X[ids], indicators(ids)] (main.py:199-204); halo rows by all-to-all when partitioned."""
        g = self.g
        if not hasattr(g, "features"):
            if differentiable and self.embed and torch.is_grad_enabled():
                return ops.GatherRowsFn.apply(self.X, ids, g.ind_code if num_ind else None, epoch, num_ind)
            return ops.gather_rows(self.X.detach(), ids, g.ind_code if num_ind else None, epoch, num_ind)
        x = g.features(ids)
        if not num_ind:
            return x
        code = g.ind_code[ids.long()]
        live = ((code >> 8) & 0xffffff) == epoch           # (int32 code: epochs >= 2^23 set the sign bit)
        shifts = torch.arange(num_ind, device=code.device, dtype=torch.int32)
        ind = (((code.unsqueeze(1) >> shifts) & 1) * live.unsqueeze(1)).to(torch.float32)
        return torch.cat([x, ind], dim=1).contiguous()

    # ------------------------------------------------------------------
    def step(self, target_nodes: torch.Tensor, uniforms_fn: Optional[Callable] = None,
             inject_logits_fn: Optional[Callable] = None, trace: bool = False, train: bool = True) -> Dict:
        g, dev, N = self.g, self.g.device, self.g.num_nodes
        hops, K, num_ind = self.hops, self.K, self.num_ind
        epoch = self.epoch = g.next_epoch()       # a fresh tag per batch replaces indicator_features.zero_() (main.py:167)
        targets = target_nodes.to(device=dev, dtype=torch.int32).contiguous()
        B = targets.numel()
        if num_ind:
            ops.indicator_mark(g.ind_code, targets, epoch, num_ind - 1)              # main.py:168
        previous = targets                                                          # main.py:163
        e_cap, n_cap = self._caps(B + K)
        src, dst, d_e = self._expand(previous, e_cap)                               # main.py:180 (hop 0)
        log_probs: List[torch.Tensor] = []
        kept_all: List[torch.Tensor] = []
        k_hop: List[tuple] = []
        agg_counts: List[torch.Tensor] = []
        hop_trace: List[Dict] = []
        all_stats: List[Dict] = []
        log_z = torch.zeros(1, device=dev)                                          # main.py:176
        use_gfn = not self.random_sampling and inject_logits_fn is None
        for hop in range(hops):                                                     # main.py:178
            # ---- frontier compaction (main.py:183-190): ascending-id batch / neighbour nodes
            if not self._marked:
                ops.bitmap_mark(g.prev_bits, None, previous, N, status=g.status)
                if self._eoff is not None:   # source endpoints: one mark per queried row that has edges
                    ops.bitmap_mark_rows(g.bits, g.bits1, previous, self._eoff, N, status=g.status)
                else:
                    ops.bitmap_mark(g.bits, g.bits1, src, N, d_n=d_e, status=g.status)
                ops.bitmap_mark(g.bits, g.bits1, dst, N, d_n=d_e, status=g.status)
            # (+ main.py:191: the new neighbours' indicator column, set by the same launch)
            batch, neigh, nbl, counts = ops.frontier_compact(g.bits, g.bits1, g.prev_bits, N, n_cap,
                                                             node_map=g.node_map, status=g.status,   # + main.py:194
                                                             ind_code=g.ind_code if num_ind else None, epoch=epoch, ind_bit=hop)
            ops.bitmap_clear(g.prev_bits, previous)
            e, nb, nn, st = torch.cat([d_e, counts, g.status]).tolist()              # the hop's one host read
            if st:
                g.check_status(f"hop {hop}")
            batch_nodes, neighbor_nodes, nb_local = batch[:nb], neigh[:nn], nbl[:nn]
            gsrc, gdst = src[:e], dst[:e]
            if trace:
                lsrc = ops.tensormap_map(g.node_map, gsrc)                           # main.py:195
                ldst = ops.tensormap_map(g.node_map, gdst)
            # (the relabel of main.py:195 happens inside the graph build: it reads the TensorMap itself)
            prep = ops.PreparedGraph(gsrc, gdst, nb, status=g.status, src_grouped=True, items_fwd=False, node_map=g.node_map)
            # ---- inclusion logits (main.py:198-213)
            if self.random_sampling:
                cand_logits = torch.full((nn, 1), 100.0, device=dev)                 # main.py:207
            elif inject_logits_fn is not None:
                cand_logits = inject_logits_fn(hop, batch_nodes).reshape(-1, 1)[nb_local.long()]
            else:
                x = self._features(batch_nodes, epoch, num_ind)                       # main.py:199-204
                node_logits, _ = self.gcn_gf(x, prep)                                # main.py:210
                agg_counts += [prep.rowptr_t[nb], prep.rowptr_t[nb]]
                cand_logits = node_logits[nb_local.long()]                           # main.py:213
            # ---- exact-k draw (main.py:216-220)
            u = self._uniforms(hop, nn, uniforms_fn) if K < nn else None
            kept, log_prob, stats = sample_neighborhoods_from_probs(cand_logits, neighbor_nodes, K, uniforms=u)
            kept_all.append(kept)                                                    # main.py:221
            if hop == 0 and use_gfn:                                                 # main.py:223-228
                xz = self._features(batch_nodes)
                pred_z = self.gcn_z(xz, prep)[0].squeeze()
                log_z = pred_z.mean() - self.log_z_init
                agg_counts += [prep.rowptr_t[nb], prep.rowptr_t[nb]]
            log_probs.append(log_prob)
            all_stats.append(stats)
            batch_next = torch.cat([targets, kept.to(torch.int32)])                  # main.py:236-238
            # ---- one expansion of batch_next serves slice_adjacency(rows=batch_next, cols=previous)
            #      (main.py:241-243) and the next hop's get_neighborhoods (main.py:180)
            ops.slice_mark(g.mult, previous)
            e_cap, n_cap = self._caps(batch_next.numel())
            src, dst, d_e = self._expand(batch_next, e_cap, mark=hop + 1 < hops)     # (the last one only feeds the slice)
            out_cap = min(e_cap, batch_next.numel() * previous.numel())
            ksrc, kdst, kcnt = ops.slice_filter(g.mult, src, dst, out_cap, d_e=d_e, status=g.status)
            ops.slice_mark(g.mult, previous, unmark=True)
            k_hop.append((ksrc, kdst, kcnt))
            if trace:
                hop_trace.append(dict(neighborhoods=torch.stack([gsrc, gdst]), batch_nodes=batch_nodes,
                                      neighbor_nodes=neighbor_nodes, local_neighborhoods=torch.stack([lsrc, ldst]),
                                      nb_local=nb_local, kept=kept, log_prob=log_prob.detach(),
                                      cand_logits=cand_logits.detach(), stats=stats,
                                      indicator_rows=(self._features(batch_nodes, epoch, num_ind)[:, self.F:]
                                                      if num_ind else None)))
            previous = batch_next                                                    # main.py:247
        # ---- final relabel (main.py:252-256): all_nodes ascending, local edge lists, classifier input
        ops.bitmap_mark(g.bits, g.bits1, targets, N, status=g.status)
        for kept in kept_all:
            ops.bitmap_mark(g.bits, g.bits1, kept.to(torch.int32), N, status=g.status)
        n_all_cap = B + hops * K + 1
        alln, _, _, counts = ops.frontier_compact(g.bits, g.bits1, None, N, n_all_cap, node_map=g.node_map,
                                                  status=g.status)
        host = torch.cat([counts[:1], g.status] + [kc for (_, _, kc) in k_hop]).tolist()   # the step's last size read
        n_all, st, kcounts = host[0], host[1], host[2:]
        if st:
            g.check_status("final relabel")
        all_nodes = alln[:n_all]
        edge_lists = []
        for (ksrc, kdst, _), m in zip(k_hop, kcounts):
            edge_lists.append((ops.tensormap_map(g.node_map, ksrc[:m]), ops.tensormap_map(g.node_map, kdst[:m]), m))
        local_targets = ops.tensormap_map(g.node_map, targets)                       # main.py:259
        out: Dict = dict(n_all=n_all, sampled_edges=[m for *_, m in edge_lists])
        if trace:
            for h, (ks, kd, _), m in zip(hop_trace, k_hop, kcounts):
                h["k_hop_edges"] = torch.stack([ks[:m], kd[:m]])
            out.update(hops=hop_trace, all_nodes=all_nodes, local_target_ids=local_targets,
                       edge_indices=[torch.stack([a, b]) for a, b, _ in edge_lists])
        if self.gcn_c is None:
            return out
        preps = [ops.PreparedGraph(a, b, n_all, status=g.status, src_grouped=True) for a, b, _ in edge_lists]
        xc = self._features(all_nodes, differentiable=True)                           # main.py:256
        logits, mem = self.gcn_c(xc, preps)                                          # main.py:257
        n_layers = len(self.gcn_c.gcn_layers)
        used = [preps[-i] for i in range(1, n_layers)] + [preps[0]]                  # gcn.py:31,35
        agg_counts += [p.rowptr_t[n_all] for p in used]
        tgt = self.y[targets.long()]
        loss_c = self.loss_fn(logits[local_targets.long()], tgt)
        if self.reg_param:
            loss_c = loss_c + self.reg_param * torch.sum(torch.var(logits, dim=1))   # main.py:260-261
        if train:
            if self.opt_c is not None:
                self.opt_c.zero_grad(set_to_none=False)     # main.py:263 (in place: a captured step may share the .grad buffers)
            loss_c.backward()                                                        # main.py:267
            if self.grad_sync is not None:
                self.grad_sync(list(self.gcn_c.parameters()) + ([self.X] if self.embed else []))
            if self.opt_c is not None:
                self._opt_step(self.opt_c)                                           # main.py:268
        out.update(loss_c=loss_c.detach(), logits=logits.detach() if trace else None, gcn_mem_alloc=mem,
                   stats=all_stats)
        if use_gfn:                                                                  # main.py:272-289
            cost = loss_c.detach()                                                   # main.py:274
            tot = torch.sum(torch.cat(log_probs, dim=0))                             # main.py:276
            if self.reinforce_baseline:
                loss_gfn = -tot * cost                                               # main.py:279
            else:
                loss_gfn = (log_z + tot + self.loss_coef * cost) ** 2                # main.py:282
            if train:
                if self.opt_gf is not None:
                    self.opt_gf.zero_grad(set_to_none=False)                         # main.py:273
                loss_gfn.backward()                                                  # main.py:287
                if self.grad_sync is not None:
                    self.grad_sync(list(self.gcn_gf.parameters()) + list(self.gcn_z.parameters()))
                if self.opt_gf is not None:
                    self._opt_step(self.opt_gf)                                      # main.py:289
            out.update(loss_gfn=loss_gfn.detach().reshape(-1)[0], log_z=log_z.detach().reshape(-1)[0],
                       tot_log_prob=tot.detach())
        out["agg_counts"] = torch.stack(agg_counts) if agg_counts else None          # device; summed lazily
        return out

    def _opt_step(self, opt):
        """opt.step().  A plain torch.optim.Adam (no amsgrad) is stepped by ONE launch over its own state tensors
        (ops.FusedAdam: the same update, state_dict() stays torch's) — torch's foreach step is ~0.3 ms of host time per call,
        a sixth of this loop; any other optimiser steps through its own .step()."""
        fused = self._fused_opt.get(id(opt))
        if fused is None:
            fused = False
            if type(opt) is torch.optim.Adam and all(not g.get("amsgrad", False) and not g.get("differentiable", False)
                                                     for g in opt.param_groups):
                try:
                    fused = ops.FusedAdam([opt])
                    fused._hyper = self._hyper(opt)
                except (ValueError, RuntimeError):
                    fused = False
            self._fused_opt[id(opt)] = fused
        if fused is False:
            opt.step()
            return
        h = self._hyper(opt)
        if h != fused._hyper:                     # an lr schedule (or the user) changed a hyper-parameter: new descriptors
            fused.refresh()
            fused._hyper = h
        fused.step()

    @staticmethod
    def _hyper(opt):
        return tuple((g["lr"], tuple(g["betas"]), g["eps"], g.get("weight_decay", 0.0), g.get("maximize", False), len(g["params"]))
                     for g in opt.param_groups)

    @staticmethod
    def edges_aggregated(step_out: Dict) -> int:
        """Σ over every GCNConv forward of its non-self-loop edge count (SURVEY §8d metric)."""
        c = step_out.get("agg_counts")
        return int(c.sum().item()) if c is not None else 0
