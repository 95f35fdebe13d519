#!/usr/bin/env python3
"""Per-kernel micro-benchmark on the shapes of the north-star workload (products-like hop 2:
n_b = 37.5k frontier rows, e = 38k edges, 512 sources, F_in = 104, H = 256).  HIP-event timing of
R back-to-back launches on torch's current stream; prints achieved GB/s or TFLOP/s against the
MI355X peaks.  Usage: python profiles/microbench.py [--reps 50]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grapes_amd import ops  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--n", type=int, default=37500)
    ap.add_argument("--m", type=int, default=512)
    ap.add_argument("--e", type=int, default=38000)
    args = ap.parse_args()
    dev = "cuda"
    n, m, e, Fi, H = args.n, args.m, args.e, 104, 256
    rng = np.random.default_rng(0)
    # frontier-shaped edge list: m sources (power-law out-degree incl. one hub), destinations ascending
    w = rng.pareto(1.2, m) + 1
    w[0] = w.sum() * 0.15
    deg = np.maximum(1, (w / w.sum() * e).astype(np.int64))
    srcs = np.sort(rng.permutation(n)[:m])
    src = np.repeat(srcs, deg)
    dst = np.concatenate([np.sort(rng.permutation(n)[:d]) for d in deg])
    e = len(src)
    ls, ld = torch.from_numpy(src).to(dev, torch.int32), torch.from_numpy(dst).to(dev, torch.int32)
    x = torch.randn(n, Fi, device=dev)
    W = torch.randn(H, Fi, device=dev) * 0.1
    w1 = torch.randn(1, H, device=dev) * 0.1
    b = torch.randn(H, device=dev)
    R = args.reps
    rows = []

    def rec(name, us, gbytes=None, gflop=None):
        s = f"{name:44s} {us:9.2f} us"
        if gbytes is not None:
            s += f"  {gbytes / us * 1e6:8.1f} GB/s ({gbytes / us * 1e6 / 8000 * 100:5.1f}% of 8 TB/s)"
        if gflop is not None:
            s += f"  {gflop / us * 1e6 / 1e3:8.2f} TFLOP/s ({gflop / us * 1e6 / 1e3 / 157.3 * 100:5.1f}% of 157.3)"
        print(s, flush=True)
        rows.append(s)

    prep = ops.PreparedGraph(ls, ld, n, src_grouped=True)
    rec("gcn_prepare (grouped)", timeit(lambda: ops.PreparedGraph(ls, ld, n, src_grouped=True), R))
    rec("gcn_prepare (generic, incl. hub sort)", timeit(lambda: ops.PreparedGraph(ls, ld, n), 3))
    h = ops.linear_fwd(x, W)
    rec(f"linear_fwd  [{n}x{Fi}]x[{Fi}x{H}]", timeit(lambda: ops.linear_fwd(x, W, out=h), R), gflop=2 * n * Fi * H / 1e9)
    out = ops.gcn_aggregate_fwd(h, prep, b, True)
    spmm_bytes = 4 * ((e + n) * H + n * H + (e + n) + (n + 1) + n + H) / 1e9
    rec(f"gcn_aggregate_fwd F={H} (n={n}, e={e})", timeit(lambda: ops.gcn_aggregate_fwd(h, prep, b, True, out=out), R), gbytes=spmm_bytes)
    dout = torch.randn(n, H, device=dev)
    rec(f"gcn_aggregate_bwd F={H} (relu+colsum+SpMMᵀ)", timeit(lambda: ops.gcn_aggregate_bwd(dout, prep, relu_out=out), R),
        gbytes=spmm_bytes + 4 * 3 * n * H / 1e9)
    dh, _ = ops.gcn_aggregate_bwd(dout, prep, relu_out=out)
    dw = torch.empty(H, Fi, device=dev)
    rec(f"linear_bwd_weight dW[{H}x{Fi}] K={n}", timeit(lambda: ops.linear_bwd_weight(dh, x, out=dw), R), gflop=2 * n * Fi * H / 1e9)
    dx = torch.empty(n, Fi, device=dev)
    rec(f"linear_bwd_input dX[{n}x{Fi}]", timeit(lambda: ops.linear_bwd_input(dh, W, out=dx), R), gflop=2 * n * Fi * H / 1e9)
    h1 = ops.linear_fwd(out, w1)
    rec(f"linear_fwd 1-wide head (gemv {n}x{H})", timeit(lambda: ops.linear_fwd(out, w1, out=h1), R), gbytes=4 * n * H / 1e9)
    o1 = ops.gcn_aggregate_fwd(h1, prep, None, False)
    rec("gcn_aggregate_fwd F=1", timeit(lambda: ops.gcn_aggregate_fwd(h1, prep, None, False, out=o1), R))
    d1 = torch.randn(n, 1, device=dev)
    rec("gcn_aggregate_bwd F=1", timeit(lambda: ops.gcn_aggregate_bwd(d1, prep), R))
    dh1, _ = ops.gcn_aggregate_bwd(d1, prep)
    rec("linear_bwd_weight 1-wide (colsum)", timeit(lambda: ops.linear_bwd_weight(dh1, out), R), gbytes=4 * n * H / 1e9)
    rec("linear_bwd_input 1-wide (outer)", timeit(lambda: ops.linear_bwd_input(dh1, w1), R), gbytes=4 * n * H / 1e9)
    # sampler
    nn_ = n - m
    logits = torch.randn(n, device=dev)
    idx = torch.arange(m, n, device=dev, dtype=torch.int32)
    u = torch.rand(nn_, device=dev)
    rec(f"gumbel_topk n={nn_} k=256", timeit(lambda: ops.gumbel_topk(logits, 256, uniforms=u, logit_index=idx), R))
    # features
    N, F = 2_449_029, 100
    X = torch.randn(N, F, device=dev)
    ids = torch.from_numpy(np.sort(rng.permutation(N)[:n])).to(dev, torch.int32)
    code = torch.zeros(N, dtype=torch.int32, device=dev)
    xo = torch.empty(n, F + 4, device=dev)
    rec(f"gather_rows {n}x({F}+4)", timeit(lambda: ops.gather_rows(X, ids, code, 1, 4, out=xo), R), gbytes=2 * 4 * n * (F + 4) / 1e9)
    # compaction at products scale
    W_ = (N + 63) // 64
    bits = torch.zeros(W_, dtype=torch.int64, device=dev)
    bits1 = torch.zeros((W_ + 63) // 64, dtype=torch.int64, device=dev)
    prevb = torch.zeros(W_, dtype=torch.int64, device=dev)
    nm = torch.empty(N, dtype=torch.int32, device=dev)

    def compact():
        ops.bitmap_mark(bits, bits1, ids, N)
        ops.frontier_compact(bits, bits1, prevb, N, n + 1024, node_map=nm)

    rec("bitmap_mark + frontier_compact (4 launches)", timeit(compact, R))
    path = os.path.join(ROOT, "gpurun_out", "microbench.txt")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    open(path, "w").write("\n".join(rows) + "\n")


if __name__ == "__main__":
    main()
