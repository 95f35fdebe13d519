#!/usr/bin/env python3
"""The one-launch compaction of a papers100M-sized frontier bitmap (111M bits = 1.74M words, ~25k bits set, ~800 previous-hop
bits) on its own: HIP-event time of the launch, bits re-set by a scatter before every call.  GRAPES_COMPACT_WIDE=0 selects the
two-launch form."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from grapes_amd import ops
dev = torch.device("cuda", 0)
N = int(os.environ.get("N", 111_059_956)); W = (N + 63) // 64
rng = np.random.default_rng(0)
ids = np.unique(rng.integers(0, N, int(os.environ.get("SET", 25_000)))); prev = np.unique(rng.integers(0, N, 800))
def words(a):
    w = torch.from_numpy((a // 64).astype(np.int64)).to(dev); m = torch.from_numpy((np.uint64(1) << (a % 64).astype(np.uint64)).view(np.int64)).to(dev)
    return w, m
wi, mi = words(ids); wp, mp = words(prev)
bits = torch.zeros(W, dtype=torch.int64, device=dev); prev_bits = torch.zeros(W, dtype=torch.int64, device=dev)
node_map = torch.full((N,), -1, dtype=torch.int32, device=dev); ind_code = torch.zeros(N, dtype=torch.int32, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
prev_bits.index_put_((wp,), mp, accumulate=True)          # (distinct ids: no carries)
n_cap = 1 << 17
ts = []
for it in range(14):
    bits.index_put_((wi,), mi, accumulate=True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = ops.frontier_compact(bits, None, prev_bits, N, n_cap, node_map=node_map, status=status, ind_code=ind_code, epoch=3, ind_bit=1,
                               want_cand_pos=True)
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
print("set bits", len(ids), "counts", out[3][:2].tolist(), "status", int(status.item()))
print("compaction launch(es), event-timed (includes ~6-8 us of launch overhead): median %.1f us, min %.1f" % (sorted(ts[3:])[len(ts[3:]) // 2], min(ts[3:])))
