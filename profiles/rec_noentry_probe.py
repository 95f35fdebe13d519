#!/usr/bin/env python3
"""VERDICT r03 item 2's probe for the record-driven 256-wide aggregation: the kernel over a graph WITHOUT entries (every row is its
unit self-loop: 1 KB in, 1 KB out per row) against a copy_ of the same bytes — what the row loop costs before any gather happens.
Also the row-per-wavefront kernel it replaced (head_local=False).  HIP-event time per launch, caches flushed between launches."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from grapes_amd import ops, _lib
_lib.load()
dev = torch.device("cuda", 0)
n, f = 77000, 256
h = torch.randn(n, f, device=dev); out = torch.empty_like(h)
st = torch.zeros(1, dtype=torch.int32, device=dev)
es = torch.tensor([0], dtype=torch.int32, device=dev); ed = torch.tensor([1], dtype=torch.int32, device=dev)     # (one edge: an empty list builds no records)
iota = torch.arange(n, dtype=torch.int32, device=dev)
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)


def timed(fn, reps=12):
    ts = []
    for _ in range(reps):
        flush.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

rec = ops.PreparedGraph(es, ed, n, status=st, src_grouped=True, items_fwd=False, head_ids=iota, head_local=True)
old = ops.PreparedGraph(es, ed, n, status=st, src_grouped=True, items_fwd=False)
assert ops._rec_form(rec, n, f) and not ops._rec_form(old, n, f)
t_copy = timed(lambda: out.copy_(h))
t_rec = timed(lambda: ops.gcn_aggregate_fwd(h, rec, None, True, out=out))
t_old = timed(lambda: ops.gcn_aggregate_fwd(h, old, None, True, out=out))
by = 2 * n * f * 4 / 1e6
print(f"{n} rows x {f}: {by:.0f} MB in + out")
print(f"copy_                         {t_copy:7.1f} us  {by / t_copy:6.2f} TB/s")
print(f"record-driven, no entries     {t_rec:7.1f} us  = {t_rec / t_copy:.2f} x copy_")
print(f"row per wavefront, no entries {t_old:7.1f} us  = {t_old / t_copy:.2f} x copy_")
