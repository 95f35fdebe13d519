#!/usr/bin/env python3
"""Phase stamps inside gemm_wsplit_f32_k (diagnostic build: make -C grapes_amd/csrc stamps; GRAPES_LIB_PATH=.../libgrapes_hip_stamps.so):
when thread 0 of the first 64 workgroups ARRIVES at each point of panel iterations 1 and 2 (no waits inserted)."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from grapes_amd import ops, _lib
lib = _lib.load()
lib.grapes_stamp_set_gemm.argtypes = [C.c_void_p]
dev = torch.device("cuda", 0)
n, K, N = 41627, 104, 256
x = torch.randn(n, K, device=dev); w = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev); hw = torch.randn(N, device=dev)
buf = torch.zeros(64 * 16, dtype=torch.int64, device=dev)
rate = float(lib.grapes_kernel_clock_rate_khz()) * 1e3
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)
print("mode:", "gate bits" if "bits" in sys.argv[1:] else "activations", "| cold input" if "warm" not in sys.argv[1:] else "| warm input")
lib.grapes_stamp_set_gemm(buf.data_ptr())
acc = []
for r in range(12):
    if "warm" not in sys.argv[1:]: flush.zero_()
    buf.zero_()
    if "bits" in sys.argv[1:]: ops.linear_relu_head_fwd_bits(x, w, b, hw.view(1, -1))
    else: ops.linear_bias_act_head_fwd(x, w, b, True, hw)
    torch.cuda.synchronize()
    acc.append(buf.cpu().numpy().reshape(64, 16).astype(np.float64))
st = np.stack(acc[2:])
names = ["top", "head combined", "MFMAs issued", "next panel staged (split + LDS)", "loads + stores issued (+ row sums)", "barrier passed"]
for it in range(2):
    print(f"panel iteration {it + 1} (us since its top, median over workgroups and launches)")
    for k in range(1, 6):
        d = ((st[:, :, it * 6 + k] - st[:, :, it * 6]) / rate * 1e6)
        ok = (st[:, :, it * 6 + k] > 0) & (st[:, :, it * 6] > 0)
        print(f"   {names[k]:40s} {np.median(d[ok]):6.2f}")
d = (st[:, :, 6] - st[:, :, 0]) / rate * 1e6
print("top of iteration 2 - top of iteration 1:", np.median(d[(st[:, :, 6] > 0) & (st[:, :, 0] > 0)]))

print("inside the store phase of iteration 1 (us since its top):")
for sl, what in ((3, "staged"), (12, "next loads issued"), (13, "accumulators read"), (14, "stores issued"), (4, "head sums done"), (5, "barrier passed")):
    d = ((st[:, :, sl] - st[:, :, 0]) / rate * 1e6); ok = (st[:, :, sl] > 0) & (st[:, :, 0] > 0)
    print(f"   {what:30s} {np.median(d[ok]):6.2f}")
