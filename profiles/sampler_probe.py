#!/usr/bin/env python3
"""Sampler draw at the step's hop-2 shape (40k batch rows, 37k candidates, k = 256) in a loop, for rocprofv3 --kernel-trace
--stats: per-kernel time of sampler_agg_keys_k / sampler_emit_k (and the unfused narrow + keys launches with --unfused).
GRAPES_KEYS_DBG (diagnosis builds only) switches parts of the key kernel off."""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grapes_amd import ops

ap = argparse.ArgumentParser(); ap.add_argument("--unfused", action="store_true"); ap.add_argument("--reps", type=int, default=300)
args = ap.parse_args()
rng = np.random.default_rng(0)
n_rows, n_cand, k = 40700, 37000, 256
cap = 131072 + 513
src = np.sort(rng.integers(0, 512, 41000)); dst = rng.integers(0, n_rows, 41000)
key = np.unique(src.astype(np.int64) * n_rows + dst); src, dst = key // n_rows, key % n_rows
keep = src != dst; src, dst = src[keep], dst[keep]
t = lambda a, d=None: (torch.as_tensor(np.ascontiguousarray(a)).to(d) if d else torch.as_tensor(np.ascontiguousarray(a))).cuda()
st = torch.zeros(1, dtype=torch.int32, device="cuda")
d_rows = torch.tensor([n_rows], dtype=torch.int32, device="cuda")
prep = ops.PreparedGraph(t(src, torch.int32), t(dst, torch.int32), cap, d_n=d_rows, status=st, src_grouped=True, items_fwd=False)
hw = t(rng.standard_normal(cap).astype(np.float32)); bias = t(np.array([0.1], np.float32))
cand_rows = np.sort(rng.permutation(n_rows)[:n_cand])
nbl = torch.zeros(cap, dtype=torch.int32, device="cuda"); nbl[:n_cand] = t(cand_rows, torch.int32)
cp = np.full(cap, -1, np.int32); cp[cand_rows] = np.arange(n_cand); cand_pos = t(cp)
ids = t(rng.permutation(10 * cap)[:cap].astype(np.int32))
d_nc = torch.tensor([n_cand], dtype=torch.int32, device="cuda")
off = torch.zeros(1, dtype=torch.int64, device="cuda")
prefix = t(np.arange(256, dtype=np.int32))
flush = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
for r in range(args.reps):
    flush.add_(1.0)                                   # 256 MB touched between draws: cold caches, like in the step
    if args.unfused:
        lg = ops.gcn_aggregate_fwd(hw.view(-1, 1), prep, bias, False)
        ops.gumbel_topk(lg.view(-1), k, logit_index=nbl, candidate_ids=ids, n=cap, d_n=d_nc, philox_seed=1, d_philox_offset=off, prefix_ids=prefix)
    else:
        ops.gumbel_topk(None, k, logit_index=nbl, candidate_ids=ids, n=cap, d_n=d_nc, philox_seed=1, d_philox_offset=off, prefix_ids=prefix,
                        agg=(hw, prep, bias, cand_pos))
torch.cuda.synchronize()
print("ok")
