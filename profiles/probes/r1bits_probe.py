#!/usr/bin/env python3
"""The by-source backward aggregation of a transform-first layer (rank-1 gated rows) on a Reddit-hop-shaped graph: n rows, S
sources of out-degree D (the sampled nodes), every other row its self-loop only.  Activation-row form vs gate-bit form,
back-to-back launches, HIP-event timed (per-launch = total / reps)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from grapes_amd import ops
dev = torch.device("cuda", 0)
n, S, D, f = int(os.environ.get("N", 79000)), int(os.environ.get("S", 768)), int(os.environ.get("D", 150)), 256
rng = np.random.default_rng(0)
srcs = np.sort(rng.choice(n, S, replace=False))
src = np.repeat(srcs, D).astype(np.int32)
dst = rng.integers(0, n, S * D).astype(np.int32)
st = torch.zeros(1, dtype=torch.int32, device=dev)
prep = ops.PreparedGraph(torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev), n, status=st, src_grouped=True, items_fwd=False)
h = torch.randn(n, f, device=dev); b1 = torch.randn(f, device=dev); w2 = torch.randn(f, device=dev); dh2 = torch.randn(n, device=dev)
out, hw, bits = ops.gcn_aggregate_fwd_head(h, prep, b1, True, w2, want_bits=True)
def timed(fn, reps=30):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
print(f"n={n} sources={S} x {D} entries, f={f}")
print("  forward aggregation + head + bits      %7.1f us" % timed(lambda: ops.gcn_aggregate_fwd_head(h, prep, b1, True, w2, want_bits=True)))
print("  forward aggregation alone              %7.1f us" % timed(lambda: ops.gcn_aggregate_fwd(h, prep, b1, True)))
print("  backward, activation rows (3 + 2 launches) %7.1f us" % timed(lambda: ops.gcn_aggregate_bwd_rank1(out, dh2, w2, prep)))
print("  backward, gate bits        (3 + 2 launches) %7.1f us" % timed(lambda: ops.gcn_aggregate_bwd_rank1(out, dh2, w2, prep, gate_bits=bits)))
x = torch.empty(n, f, device=dev)
print("  torch fill of the output               %7.1f us" % timed(lambda: x.fill_(1.0)))
print("  torch copy act -> output               %7.1f us" % timed(lambda: x.copy_(out)))
if os.environ.get("COLD"):
    big = torch.empty(1 << 28, device=dev)
    for _ in range(10):
        big.zero_(); ops.gcn_aggregate_bwd_rank1(out, dh2, w2, prep, gate_bits=bits)
        big.zero_(); ops.gcn_aggregate_bwd_rank1(out, dh2, w2, prep)
        big.zero_(); ops.gcn_aggregate_fwd_head(h, prep, b1, True, w2, want_bits=True)
    torch.cuda.synchronize()
