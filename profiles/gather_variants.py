#!/usr/bin/env python3
"""Times the fused gather-SpMM on a real products-scale hop-2 frontier: CSR-walking kernel vs head-record kernel."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from grapes_amd import ops, synth
from grapes_amd.graph import DeviceGraph
dev = torch.device("cuda", 0)
N, deg, maxdeg, F, C, B, K, hops = synth.CONFIGS["products"]
rowptr, col = synth.synth_graph_device(N, deg, maxdeg, seed=0, device=dev)
g = DeviceGraph(rowptr, col, N)
gen = torch.Generator(device=dev); gen.manual_seed(1)
X = torch.randn(N, F, device=dev, generator=gen)
prev = torch.randperm(N, device=dev, generator=gen)[:512].to(torch.int32)
e_cap = 1 << 17; n_cap = e_cap + 513
eoff, d_e = ops.frontier_offsets(g.rowptr, prev)
src, dst, _ = ops.frontier_expand(g.rowptr, g.col, prev, eoff, e_cap, status=g.status)
ops.bitmap_mark(g.prev_bits, None, prev, N, status=g.status)
ops.bitmap_mark_rows(g.bits, g.bits1, prev, eoff, N, status=g.status)
ops.bitmap_mark(g.bits, g.bits1, dst, N, d_n=d_e, status=g.status)
batch, neigh, nbl, counts = ops.frontier_compact(g.bits, g.bits1, g.prev_bits, N, n_cap, node_map=g.node_map, status=g.status)
d_nb = counts[0:1]
ep = torch.ones(1, dtype=torch.int32, device=dev)
ops.indicator_mark(g.ind_code, neigh, 0, 2, d_n=counts[1:2], d_epoch=ep)
plain = ops.PreparedGraph(src, dst, n_cap, d_n=d_nb, d_e=d_e, status=g.status, src_grouped=True, items_fwd=False, node_map=g.node_map)
heads = ops.PreparedGraph(src, dst, n_cap, d_n=d_nb, d_e=d_e, status=g.status, src_grouped=True, items_fwd=False, node_map=g.node_map, head_ids=batch)
n, e = int(d_nb), int(d_e)
print(f"frontier rows {n}, edges {e}")
def run(prep, reps=100):
    for _ in range(5): ops.gcn_aggregate_gather(X, batch, prep, g.ind_code, 0, 4, d_epoch=ep)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = torch.empty((n_cap, F + 4), device=dev)
    a.record()
    for _ in range(reps): ops.gcn_aggregate_gather(X, batch, prep, g.ind_code, 0, 4, d_epoch=ep, out=out)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
alg = 4 * ((e + n) * 104 + n * 104 + (e + n) + (n + 1) + n + 104)
for name, prep in (("csr-walk", plain), ("row-heads", heads)):
    t = run(prep)
    print(f"{name:10s} {t:7.2f} us  -> {alg / t / 1e6:7.1f} GB/s algorithmic")
xg = torch.empty((n, 104), device=dev)
def run_copy(reps=100):
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): ops.gather_rows(X, batch, g.ind_code, 0, 4, d_n=d_nb, d_epoch=ep)
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps * 1e3
print(f"plain row gather of the same {n} rows (no aggregation): {run_copy():.2f} us")
