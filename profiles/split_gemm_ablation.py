#!/usr/bin/env python3
"""Ablation of the split-bf16 forward GEMM: which phase bounds it?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grapes_amd import _lib
lib = _lib.load_diag()
st = torch.cuda.current_stream().cuda_stream
def run(x, w, out, dbg, reps=50):
    n, fi = x.shape; fo = w.shape[0]
    for _ in range(5): lib.grapes_debug_gemm_fwd(x.data_ptr(), w.data_ptr(), out.data_ptr(), n, fi, fo, dbg, st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): lib.grapes_debug_gemm_fwd(x.data_ptr(), w.data_ptr(), out.data_ptr(), n, fi, fo, dbg, st)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for n in (64, 8192):
    x = torch.randn(n, 104, device="cuda"); w = torch.randn(256, 104, device="cuda") * 0.1; out = torch.empty(n, 256, device="cuda")
    for bits, name in ((0, "full"), (2048, "no W loads"), (2048 | 256 | 512 | 1024, "no W loads, loads + barriers only"), (16 - 64, "fp32 W-stationary")):
        print(f"n={n} {name:28s} {run(x, w, out, 64 + bits):8.2f} us", flush=True)
for n in (131072, 37500):
    x = torch.randn(n, 104, device="cuda"); w = torch.randn(256, 104, device="cuda") * 0.1; out = torch.empty(n, 256, device="cuda")
    for bits, name in ((0, "full"), (256, "no stores"), (512, "no MFMAs"), (1024, "no staging"), (256 | 512, "no stores, no MFMAs"),
                       (512 | 1024, "no MFMAs, no staging"), (256 | 1024, "no stores, no staging"), (256 | 512 | 1024, "loads + barriers only")):
        print(f"n={n} {name:28s} {run(x, w, out, 64 | bits):8.2f} us", flush=True)
