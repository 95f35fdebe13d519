"""The reference's training iteration, restated over the DROP-IN modules exactly as INTEGRATION.md §2 leaves it: reference
main.py:157-291 with three import lines changed (`GCN`, `TensorMap`, `get_neighborhoods`, `sample_neighborhoods_from_probs`,
`slice_adjacency` from grapes_amd.modules) and NOTHING ELSE — the O(N) boolean masks stay on the host (main.py:138-140,
183-190), `data.x` and the indicator matrix stay CPU tensors that are indexed and copied to the device per hop
(main.py:199-204,227,256), index tensors are int64 on the CPU, the kept ids come back through a D2H read (utils.py:60), the
losses are read with `.item()` (main.py:269,291).

This is NOT the product path (that is step_graph.GraphedTrainer, or step.GrapesTrainer for an eager loop with everything
resident).  It exists so that the cost of "staying drop-in" has a measurement behind it (bench.py:
`config.reference_shaped_dropin_loop`) and a parity test (tests/test_hip_parity.py): a user of the reference who changes the
imports gets THIS loop.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .modules.utils import TensorMap, get_neighborhoods, sample_neighborhoods_from_probs, slice_adjacency


class ReferenceShapedLoop:
    def __init__(self, adjacency, x_cpu: torch.Tensor, y: torch.Tensor, gcn_c: nn.Module, gcn_gf: nn.Module, gcn_z: nn.Module, *,
                 sampling_hops: int = 2, num_samples: int = 16, use_indicators: bool = True, loss_coef: float = 1e4,
                 log_z_init: float = 0.0, reg_param: float = 0.0, optimizer_c=None, optimizer_gf=None, device="cuda"):
        if x_cpu.is_cuda:
            raise ValueError("the reference keeps data.x on the host (main.py:66,199-204): pass the CPU tensor")
        self.adjacency, self.x, self.y = adjacency, x_cpu, y.cpu()
        self.gcn_c, self.gcn_gf, self.gcn_z = gcn_c, gcn_gf, gcn_z
        self.hops, self.k, self.coef, self.log_z_init, self.reg = sampling_hops, num_samples, loss_coef, log_z_init, reg_param
        self.opt_c, self.opt_gf, self.device = optimizer_c, optimizer_gf, torch.device(device)
        n = x_cpu.shape[0]
        self.use_indicators = use_indicators
        self.node_map = TensorMap(size=n)                                           # main.py:102
        width = sampling_hops + 1 if use_indicators else 0                           # main.py:104-107
        self.prev_mask = torch.zeros(n, dtype=torch.bool)                            # main.py:138
        self.batch_mask = torch.zeros(n, dtype=torch.bool)                           # main.py:139
        self.indicators = torch.zeros((n, width))                                    # main.py:140
        self.loss_fn = nn.CrossEntropyLoss() if self.y.dim() == 1 else nn.BCEWithLogitsLoss()   # main.py:120-123

    def step(self, target_nodes: torch.Tensor, uniforms_fn=None) -> Dict:
        """One mini-batch.  target_nodes: int64 on the CPU, as a DataLoader over train_idx yields them (main.py:126,161)."""
        dev, nm = self.device, self.node_map
        target_nodes = target_nodes.cpu().long()
        previous = target_nodes.clone()                                              # main.py:163
        seen = torch.zeros_like(self.prev_mask)                                      # main.py:164-165
        seen[target_nodes] = True
        self.indicators.zero_()                                                      # main.py:167-168
        if self.use_indicators:
            self.indicators[target_nodes, -1] = 1.0
        hop_edges: List[torch.Tensor] = []
        log_probs: List[torch.Tensor] = []
        kept_per_hop: List[torch.Tensor] = []
        log_z = torch.tensor([0.0])                                                  # main.py:176
        for hop in range(self.hops):                                                 # main.py:178
            nbh = get_neighborhoods(previous, self.adjacency)                        # main.py:180  (int64 [2, e] on the CPU)
            self.prev_mask.zero_(); self.batch_mask.zero_()                          # main.py:183-187
            self.prev_mask[previous] = True
            self.batch_mask[nbh.view(-1)] = True
            fresh = self.batch_mask & ~self.prev_mask
            batch_nodes = nm.values[self.batch_mask]                                 # main.py:189-190 (ascending ids)
            neighbor_nodes = nm.values[fresh]
            if self.use_indicators:
                self.indicators[neighbor_nodes, hop] = 1.0                           # main.py:191
            nm.update(batch_nodes)                                                   # main.py:194-195
            local = nm.map(nbh).to(dev)
            if self.use_indicators:                                                  # main.py:198-204 (gather on the host, H2D)
                x = torch.cat([self.x[batch_nodes], self.indicators[batch_nodes]], dim=1).to(dev)
            else:
                x = self.x[batch_nodes].to(dev)
            logits, _ = self.gcn_gf(x, local)                                        # main.py:210
            logits = logits[nm.map(neighbor_nodes)]                                  # main.py:213
            u = uniforms_fn(hop, neighbor_nodes.numel()) if (uniforms_fn is not None and self.k < neighbor_nodes.numel()) else None
            kept, log_prob, _stats = sample_neighborhoods_from_probs(logits, neighbor_nodes, self.k, uniforms=u)   # main.py:216-220
            seen[kept] = True                                                        # main.py:221
            if hop == 0:                                                             # main.py:223-228
                pred = self.gcn_z(self.x[batch_nodes].to(dev), local)[0].squeeze()
                log_z = pred.mean() - self.log_z_init
            log_probs.append(log_prob)
            kept_per_hop.append(kept)
            batch_next = torch.cat([target_nodes, kept], dim=0)                      # main.py:236-238
            hop_edges.append(slice_adjacency(self.adjacency, rows=batch_next, cols=previous))   # main.py:241-244
            previous = batch_next.clone()                                            # main.py:247
        all_nodes = nm.values[seen]                                                  # main.py:252-256
        nm.update(all_nodes)
        edge_indices = [nm.map(e).to(dev) for e in hop_edges]
        x = self.x[all_nodes].to(dev)
        logits, _ = self.gcn_c(x, edge_indices)                                      # main.py:257
        local_targets = nm.map(target_nodes)                                         # main.py:259
        loss_c = self.loss_fn(logits[local_targets], self.y[target_nodes].to(dev)) \
            + self.reg * torch.sum(torch.var(logits, dim=1))                         # main.py:260-261
        if self.opt_c is not None:
            self.opt_c.zero_grad()                                                   # main.py:263
        loss_c.backward()                                                            # main.py:267
        if self.opt_c is not None:
            self.opt_c.step()                                                        # main.py:268
        batch_loss_c = loss_c.item()                                                 # main.py:269
        cost = loss_c.detach()                                                       # main.py:272-282
        if self.opt_gf is not None:
            self.opt_gf.zero_grad()
        loss_gfn = (log_z.to(dev) + torch.sum(torch.cat(log_probs, dim=0)) + self.coef * cost) ** 2
        loss_gfn.backward()                                                          # main.py:287
        if self.opt_gf is not None:
            self.opt_gf.step()                                                       # main.py:289
        batch_loss_gfn = loss_gfn.item()                                             # main.py:291
        return dict(loss_c=batch_loss_c, loss_gfn=batch_loss_gfn, kept=kept_per_hop, all_nodes=all_nodes, logits=logits.detach(),
                    edges=[int(e.shape[1]) for e in edge_indices])
