#!/usr/bin/env python3
"""Ablation of the gathered-operand forward GEMM on the bf16 pipe (gemm_tsplit_fwd_k) at Reddit's hop-1 shape (77k rows x 608 -> 256):
which phase bounds a K step — the MFMAs, the HBM gather of the rows, the W image loads, the split + LDS staging?"""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grapes_amd import ops, _lib
lib = _lib.load_diag()
lib.grapes_debug_tsplit_fwd.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
N, F, H = 232965, 604, 256
X = torch.randn(N, F, device="cuda")
w = (torch.randn(H, F, device="cuda") * 0.05).contiguous()
img = ops.weight_split_image(w)
st = torch.cuda.current_stream().cuda_stream
flush = torch.empty(96 << 20, dtype=torch.float32, device="cuda")
for n in (77000, 25000):
    ids = torch.from_numpy(np.sort(np.random.default_rng(0).permutation(N)[:n]).astype(np.int32)).cuda()
    out = torch.empty(n, H, device="cuda")
    def run(dbg, reps=20):
        ts = []
        for r in range(reps + 3):
            flush.add_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); rc = lib.grapes_debug_tsplit_fwd(X.data_ptr(), F, F, ids.data_ptr(), img.data_ptr(), out.data_ptr(), n, H, dbg, st); b.record()
            assert rc == 0, rc
            torch.cuda.synchronize()
            if r >= 3: ts.append(a.elapsed_time(b) * 1e3)
        return float(np.median(ts))
    print(f"n = {n}  (tiles of 128 rows: {(n + 127) // 128}, K steps: {(F + 31) // 32})")
    for dbg, name in ((0, "full (lockstep kernel)"), (16, "producer / consumer kernel"), (1, "no MFMAs"), (2, "gathered rows -> row 0 (no HBM gather)"),
                      (4, "one W block (no W traffic)"), (8, "no split + staging"), (2 | 4, "no gather, no W traffic"), (1 | 8, "loads only (no MFMAs, no staging)"),
                      (1 | 2 | 4, "staging only"), (2 | 4 | 8, "MFMAs only")):
        print(f"  dbg={dbg:2d} {name:44s} {run(dbg):8.1f} us", flush=True)

# ---- the weight-gradient kernel (gemm_tsplit_dw_k: producer / consumer wavefronts, split-K over the rows)
lib.grapes_debug_tsplit_dw.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
wsb = int(lib.grapes_linear_bwd_weight_gathered_split_workspace_bytes(608, H))
ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
for n in (77000, 25000):
    ids = torch.from_numpy(np.sort(np.random.default_rng(0).permutation(N)[:n]).astype(np.int32)).cuda()
    dh = torch.randn(n, H, device="cuda")
    def run(dbg, reps=20):
        ts = []
        for r in range(reps + 3):
            flush.add_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); rc = lib.grapes_debug_tsplit_dw(dh.data_ptr(), X.data_ptr(), F, F, ids.data_ptr(), n, H, ws.data_ptr(), dbg, st); b.record()
            assert rc == 0, rc
            torch.cuda.synchronize()
            if r >= 3: ts.append(a.elapsed_time(b) * 1e3)
        return float(np.median(ts))
    print(f"dW  n = {n}")
    for dbg, name in ((0, "full"), (1, "no MFMAs (producers only)"), (2, "gathered rows -> row 0"), (8, "no split + staging (loads + MFMAs)"),
                      (1 | 8, "loads only"), (1 | 2, "staging only (no HBM gather, no MFMAs)"), (2 | 8, "MFMAs only")):
        print(f"  dbg={dbg:2d} {name:44s} {run(dbg):8.1f} us", flush=True)
