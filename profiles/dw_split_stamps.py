#!/usr/bin/env python3
"""Phase stamps inside gemm_dw_split_k<true> (the sampler nets' weight-gradient GEMM, the longest launch of the products step) at the
shapes the step gives it.  Diagnostic build only:

    make -C grapes_amd/csrc stamps
    GRAPES_DIAG=1 GRAPES_LIB_PATH=grapes_amd/libgrapes_hip_stamps.so python profiles/dw_split_stamps.py

The real step (eager launches, one-graph form) runs a few batches; around the weight-gradient call the stamp table is cleared, the
device drained and the table read back.  Thread 0 (wavefront 0: stages the mask) and thread 256 (wavefront 4: stages rs * x) of the
first 64 workgroups write the 100 MHz clock at: start, and — in the loop's 7th half-iteration (it loads chunk 6, multiplies chunk 4,
stages chunk 5) — top of the iteration, loads issued, MFMAs issued, MFMAs complete, staging issued, barrier passed; then loop end
(two stamps) and kernel end.  The stamps themselves cost: the stamped iteration reads ~2.8 us where the average one is ~1.9."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], "--workload", "products"]
import bench
from grapes_amd import ops, _lib
lib = _lib.load()
lib.grapes_stamp_set_gemm.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
b = bench.Bench(bench.parse(), 1, 0, dev)
tr, g, models = b.make("single", capture=False, pipeline=False)
tr.attach_loader(b.train_idx, stride=1, offset=0)
for _ in range(4):
    tr.step_next()
torch.cuda.synchronize()
us = 1e6 / (lib.grapes_kernel_clock_rate_khz() * 1e3)
buf = torch.zeros(64 * 16, dtype=torch.int64, device=dev)
assert lib.grapes_stamp_set_gemm(buf.data_ptr()) == 0
tabs, rows = [], []
for name in ("linear_bwd_weight_bits_pair", "linear_bwd_weight_bits_multi"):
    orig = getattr(ops, name)

    def f(*a, _orig=orig, **k):
        torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
        r = _orig(*a, **k)
        torch.cuda.synchronize()
        tabs.append(buf.cpu().numpy().reshape(64, 16).astype(np.float64).copy())
        rows.append([int(t.item()) for t in a[3]])
        return r
    setattr(ops, name, f)
for _ in range(10):
    tr.step_next()
torch.cuda.synchronize()
tr.check()
st = np.stack(tabs[2:])
print(f"# {len(tabs)} calls; live rows of the row sets (last call): {rows[-1]} -> {sum(rows[-1])} rows")
W0 = [(0, "start"), (3, "chunk 4: top of the iteration"), (4, "chunk 4: loads of chunk 6 issued"),
      (5, "chunk 4: MFMAs issued"), (6, "chunk 4: MFMAs complete"), (7, "chunk 4: staging of chunk 5 issued"), (8, "chunk 4: barrier passed"),
      (2, "loop end"), (1, "kernel end (slab stored)")]
W4 = [(9, "chunk 4: top of the iteration"), (10, "chunk 4: loads issued"), (11, "chunk 4: MFMAs issued"), (12, "chunk 4: MFMAs complete"),
      (13, "chunk 4: staging issued"), (14, "chunk 4: barrier passed")]
t0 = np.where(st[:, :, 0] > 0, st[:, :, 0], np.inf).min(axis=1)
for title, table in (("wavefront 0 (mask role)", W0), ("wavefront 4 (rs * x role)", W4)):
    print(title)
    for sl, what in table:
        ok = st[:, :, sl] > 0
        d = (st[:, :, sl] - t0[:, None]) * us
        print(f"   [{sl:2d}] {what:44s} median {np.median(d[ok]):7.2f} us   p10 {np.percentile(d[ok], 10):7.2f}   p90 {np.percentile(d[ok], 90):7.2f}   ({int(ok.sum())} stamps)")
ok = (st[:, :, 3] > 0) & (st[:, :, 8] > 0)
print(f"chunk 4, wavefront 0: top -> barrier passed: median {np.median(((st[:, :, 8] - st[:, :, 3]) * us)[ok]):.2f} us")
ok = (st[:, :, 2] > 0) & (st[:, :, 0] > 0)
print(f"start -> loop end: median {np.median(((st[:, :, 2] - st[:, :, 0]) * us)[ok]):.2f} us; rows per workgroup {sum(rows[-1]) / 224:.0f} = {sum(rows[-1]) / 224 / 32:.1f} chunks")
