#!/usr/bin/env python3
"""Per-node floor of a replayed hipGraph vs eager stream launches: N dependent tiny kernels (1-element fill)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from grapes_amd import ops
x = torch.zeros(64, device="cuda")
def body(n):
    for _ in range(n): ops.fill(x, 1.0)
for n in (50, 200):
    body(n); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): body(n)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / 50 / n * 1e6
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); body(n * 5); b.record(); torch.cuda.synchronize()
    te = a.elapsed_time(b) / (n * 5) * 1e3
    print(f"{n} dependent 1-element kernels: graph replay {tg:.2f} us per node, eager stream {te:.2f} us per launch")
