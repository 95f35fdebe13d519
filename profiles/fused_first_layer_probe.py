#!/usr/bin/env python3
"""Pricing the fused first layer (gather-aggregate -> LDS panel -> MFMA -> bias/ReLU -> head)
with MEASUREMENTS of its two halves under the budgets they would have inside one workgroup:
  * the forward split GEMM (gemm_wsplit_f32_k, the gate-bit form of the step) compiled for a 768-thread workgroup
    (8 GEMM + 4 gather wavefronts -> 168 VGPRs per wavefront: `make -C grapes_amd/csrc lb768`, loaded via GRAPES_LIB_PATH);
  * the gather-SpMM (gcn_aggregate_gather_head5_k) restricted to 4 wavefronts per compute unit (one 256-thread workgroup per CU:
    grid = 256 resident workgroups, the rest of the CU's lanes idle as they would be next to 8 GEMM wavefronts).
usage (GPU box): bash profiles/fused_first_layer_probe.sh"""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grapes_amd import ops, _lib
lib = _lib.load()
N, F, IND, H = 2_449_029, 100, 4, 256
n_rows, m_src, e_target = 41600, 512, 42400                      # the step's hop-1 shape
rng = np.random.default_rng(0)
w = rng.pareto(1.2, m_src) + 1
deg = np.minimum(n_rows, np.maximum(1, (w / w.sum() * e_target).astype(np.int64)))
srcs = np.sort(rng.permutation(n_rows)[:m_src])
src = np.repeat(srcs, deg); dst = np.concatenate([np.sort(rng.permutation(n_rows)[:d]) for d in deg])
ls, ld = torch.from_numpy(src).to("cuda", torch.int32), torch.from_numpy(dst).to("cuda", torch.int32)
ids = torch.from_numpy(np.sort(rng.permutation(N)[:n_rows])).to("cuda", torch.int32)
prep = ops.PreparedGraph(ls, ld, n_rows, src_grouped=True, items_fwd=False, head_ids=ids)
X = torch.randn(N, F, device="cuda"); code = torch.zeros(N, dtype=torch.int32, device="cuda")
ax = torch.empty(n_rows, F + IND, device="cuda")
W1 = (torch.randn(H, F + IND, device="cuda") * 0.1).contiguous(); b1 = torch.zeros(H, device="cuda"); w2 = torch.randn(1, H, device="cuda") * 0.1
flush = torch.empty(96 << 20, dtype=torch.float32, device="cuda")

def timed(fn, reps=30, cold=False):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        if cold: flush.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))

res = dict(lib=os.path.basename(_lib.LIB_PATH), shape=dict(n=n_rows, e=int(prep.rowptr_t[n_rows]), K=F + IND, H=H))
res["gemm_gate_bits_us"] = timed(lambda: ops.linear_relu_head_fwd_bits(ax, W1, b1, w2))
res["gemm_activation_tile_us"] = timed(lambda: ops.linear_bias_act_head_fwd(ax, W1, b1, True, w2))
res["gather_us_warm"] = timed(lambda: ops.gcn_aggregate_gather(X, ids, prep, code, 1, IND, out=ax))
res["gather_us_cold"] = timed(lambda: ops.gcn_aggregate_gather(X, ids, prep, code, 1, IND, out=ax), cold=True)
res["GRAPES_GATHER_GRID"] = os.environ.get("GRAPES_GATHER_GRID", "default (1536 resident workgroups of 256 threads)")
print(json.dumps(res))
