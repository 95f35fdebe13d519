#!/usr/bin/env python3
"""Ablation of the forward fp32-MFMA GEMM (37.5k x 104 x 256): which phase bounds it?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grapes_amd import ops, _lib
lib = _lib.load_diag()
n, fi, fo = 37500, 104, 256
x = torch.randn(n, fi, device="cuda"); w = torch.randn(fo, fi, device="cuda") * 0.1; out = torch.empty(n, fo, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run(dbg, reps=50):
    for _ in range(5): lib.grapes_debug_gemm_fwd(x.data_ptr(), w.data_ptr(), out.data_ptr(), n, fi, fo, dbg, st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): lib.grapes_debug_gemm_fwd(x.data_ptr(), w.data_ptr(), out.data_ptr(), n, fi, fo, dbg, st)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for dbg, name in ((0, "full"), (1, "no stores"), (2, "no operand reloads"), (4, "no MFMA"), (3, "no stores, no reloads"), (5, "no stores, no MFMA"), (6, "no reloads, no MFMA"), (7, "only LDS traffic + barriers"), (16, "W-stationary kernel")):
    print(f"dbg={dbg} {name:32s} {run(dbg):8.2f} us", flush=True)
for nn in (12700, 37500, 131072):
    n = nn
    x = torch.randn(n, fi, device="cuda"); out = torch.empty(n, fo, device="cuda")
    print(f"n={n}: tiled {run(0):7.2f} us   W-stationary {run(16):7.2f} us   ({2*n*fi*fo/1e12:.4f} TFLOP)", flush=True)
