#!/usr/bin/env python3
"""How fast does the chip WRITE a fresh n x 256 fp32 matrix (the output of the by-source backward aggregation)?  torch fill /
copy of the same bytes, back to back, HIP-event timed."""
import torch
dev = torch.device("cuda", 0)
for n in (19000, 79000, 200000):
    x = torch.empty(n, 256, device=dev); y = torch.randn(n, 256, device=dev)
    big = torch.empty(1 << 28, device=dev)
    for name, fn in (("fill", lambda: x.fill_(1.0)), ("copy", lambda: x.copy_(y)), ("fill after 1 GiB flush", None)):
        ts = []
        for it in range(12):
            if fn is None:
                big.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); (fn or (lambda: x.fill_(1.0)))(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        t = sorted(ts[2:])[len(ts[2:]) // 2]
        mb = n * 1024 / 1e6 * (2 if name == "copy" else 1)
        print(f"n={n:7d} {name:24s} {t:7.1f} us  {mb / t:6.2f} TB/s (event-timed single launch: ~6-9 us of launch overhead included)")
