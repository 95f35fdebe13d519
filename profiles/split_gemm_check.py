#!/usr/bin/env python3
"""Forward GEMM of the aggregate-first layers: fp32-MFMA kernels vs the split-bf16 kernel — error against fp64 and time."""
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
torch.manual_seed(0)
for n, fi, fo, scale in ((37500, 104, 256, 1.0), (12700, 104, 256, 1.0), (131072, 104, 256, 1.0), (5000, 100, 256, 1e3), (4099, 128, 256, 1e-3), (333, 64, 96, 1.0)):
    x = torch.randn(n, fi, device="cuda") * scale; w = torch.randn(fo, fi, device="cuda") * 0.1
    ref = x.double() @ w.double().T
    mag = (x.double().abs() @ w.double().abs().T)          # sum |a.b| per output
    line = f"n={n} K={fi} N={fo} scale={scale:g}:"
    for dbg, name in ((0, "tiled fp32"), (16, "W-stationary fp32"), (64, "split bf16x3")):
        out = torch.full((n, fo), float("nan"), device="cuda")
        t = run(x, w, out, dbg)
        err = ((out.double() - ref).abs() / mag).max().item()
        rms = (((out.double() - ref) / mag) ** 2).mean().sqrt().item()
        line += f"  [{name}: {t:6.2f} us, max err/sum|ab| {err:.2e}, rms {rms:.2e}]"
    print(line, flush=True)
