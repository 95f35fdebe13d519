#!/usr/bin/env python3
"""Phase stamps inside the sampler kernels (diagnostic build: make -C grapes_amd/csrc stamps; run with
GRAPES_LIB_PATH=grapes_amd/libgrapes_hip_stamps.so).  Prints, per stamp slot, the median time since slot 0 over the first
workgroups of sampler_keys_k and sampler_emit_k at the step's hop-2 shape, with cold caches between draws."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grapes_amd import ops, _lib
lib = _lib.load()
lib.grapes_stamp_set_sampler.argtypes = [ctypes.c_void_p]
rng = np.random.default_rng(0)
n_cand, k, cap = 37000, 256, 131072 + 513
logits = torch.randn(cap, device="cuda")
nbl = torch.from_numpy(np.sort(rng.permutation(cap)[:cap]).astype(np.int32)).cuda()
ids = torch.arange(cap, dtype=torch.int32, device="cuda")
d_nc = torch.tensor([n_cand], dtype=torch.int32, device="cuda")
off = torch.zeros(1, dtype=torch.int64, device="cuda")
prefix = torch.arange(256, dtype=torch.int32, device="cuda")
flush = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
buf = torch.zeros(64 * 16, dtype=torch.int64, device="cuda")
rate = lib.grapes_kernel_clock_rate_khz() * 1e3

def run():
    res = []
    for r in range(40):
        flush.add_(1.0)
        buf.zero_()
        ops.gumbel_topk(logits, k, logit_index=nbl, candidate_ids=ids, n=cap, d_n=d_nc, philox_seed=1, d_philox_offset=off, prefix_ids=prefix)
        torch.cuda.synchronize()
        res.append(buf.cpu().numpy().reshape(64, 16).copy())
    return np.stack(res).astype(np.float64)

assert lib.grapes_stamp_set_sampler(buf.data_ptr()) == 0
st = run()
us = 1e6 / rate
nbk = min(64, (n_cand + 1023) // 1024)
def show(title, ref, slots, nb):
    print(title)
    for sl, what in slots:
        ok = (st[:, :nb, sl] > 0) & (st[:, :nb, ref] > 0)
        d = ((st[:, :nb, sl] - st[:, :nb, ref]) * us)[ok]
        if d.size == 0:
            print(f"  {what:34s} (not stamped on this path)"); continue
        print(f"  {what:34s} median {np.median(d):7.2f} us   p90 {np.percentile(d, 90):7.2f}   max {d.max():7.2f}")
show("sampler_keys_k (since its own start, first workgroups)", 11, [(12, "loop done (loads, math, LDS hist)"), (13, "hist row stored"), (14, "end (statistics partial)")], nbk)
show("sampler_emit_k (since its own start)", 0, [(8, "histogram rows summed"), (9, "selected bin collected"), (10, "passes 2-4 done"), (1, "selection returned"),
     (2, "prefix recount done"), (3, "outputs written"), (4, "ticket taken")], nbk)
ok = st[:, 0, 5] > 0
print("last workgroup's finalise (slot 5 - its slot 4): n/a per block; keys end -> emit start gap (block 0):",
      np.median((st[:, 0, 0] - st[:, 0, 14]) * us), "us")
first = st[:, :nbk, 11].min(axis=1); last_k = st[:, :nbk, 14].max(axis=1); e0 = st[:, :nbk, 0].min(axis=1); e4 = st[:, :nbk, 4].max(axis=1)
print(f"spans (median over draws): keys first-start -> last-end {np.median((last_k - first) * us):.2f} us; keys end -> emit first start {np.median((e0 - last_k) * us):.2f} us; emit first start -> last ticket {np.median((e4 - e0) * us):.2f} us")
