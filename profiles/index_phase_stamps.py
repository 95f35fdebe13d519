#!/usr/bin/env python3
"""Phase stamps INSIDE the index chain of the products step: the draw, the one-launch expansion and the
compaction at the shapes the step gives them.  Diagnostic build only:

    make -C grapes_amd/csrc stamps
    GRAPES_DIAG=1 GRAPES_LIB_PATH=grapes_amd/libgrapes_hip_stamps.so python profiles/index_phase_stamps.py [--workload products]

The REAL step (GraphedTrainer, eager launches, one-graph form) runs a few batches; around every call of the three entry points the
stamp table is cleared, the device drained and the table read back.  Thread 0 of the first 64 workgroups of a launch writes the
100 MHz wall clock at named points (GRAPES_STAMP: after waiting for its outstanding memory operations; GRAPES_STAMP_NW: on arrival).
Printed per call position of the step (hop 0 / 1 / 2, ...): median over batches and workgroups of the time since the EARLIEST
start stamp of the launch, and of the launch's span (earliest start -> latest last stamp).  Eager launches start with the caches
as the previous launch left them (nothing is flushed), like the nodes of the replayed graph."""
import argparse, ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="products")
ap.add_argument("--batches", type=int, default=12)
ap.add_argument("--per_wg", action="store_true", help="also print the compaction stamps of every stamped workgroup (hop 1)")
pa = ap.parse_args()
sys.argv = [sys.argv[0], "--workload", pa.workload]
import bench
from grapes_amd import ops, _lib
lib = _lib.load()
for name in ("grapes_stamp_set_sampler", "grapes_stamp_set_spmm"):
    getattr(lib, name).argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
args = bench.parse()
b = bench.Bench(args, 1, 0, dev)
tr, g, models = b.make("single", capture=False, pipeline=False)
tr.attach_loader(b.train_idx, stride=1, offset=0)
for _ in range(4):
    tr.step_next()
torch.cuda.synchronize()
rate = lib.grapes_kernel_clock_rate_khz() * 1e3
us = 1e6 / rate
buf_s = torch.zeros(64 * 16, dtype=torch.int64, device=dev)
buf_h = torch.zeros(64 * 16, dtype=torch.int64, device=dev)
assert lib.grapes_stamp_set_sampler(buf_s.data_ptr()) == 0
assert lib.grapes_stamp_set_spmm(buf_h.data_ptr()) == 0
records = {}        # (entry point, call index inside the step) -> list of [64, 16] tables
calls = {}


def wrap(name, buf):
    orig = getattr(ops, name)

    def f(*a, **k):
        torch.cuda.synchronize()
        buf.zero_()
        torch.cuda.synchronize()
        r = orig(*a, **k)
        torch.cuda.synchronize()
        i = calls.get(name, 0)
        calls[name] = i + 1
        records.setdefault((name, i), []).append(buf.cpu().numpy().reshape(64, 16).astype(np.float64).copy())
        return r
    setattr(ops, name, f)


wrap("gumbel_topk", buf_s)
wrap("frontier_expand_fused", buf_h)
wrap("frontier_compact", buf_h)
for _ in range(pa.batches):
    calls.clear()
    tr.step_next()
torch.cuda.synchronize()
tr.check()

SLOTS = {
    "gumbel_topk": ([0, 7, 8, 9, 10, 1, 2, 3, 11, 12, 5, 6], {0: "start", 7: "A: candidate count / index / Philox counter arrived", 8: "A: logit arrived, key computed",
                                          9: "A: workgroup's 12-bit table complete (LDS)", 10: "A: own cut found (scan of the table)",
                                          1: "A: own list + (cut, count) stored", 2: "barrier passed",
                                          3: "B: pairs + lists arrived (1 trip), table of listed keys, bin, per-workgroup counts, offsets",
                                          11: "B: in-bin keys gathered in LDS", 12: "B: ranked (512 threads share nc x nc), threshold known",
                                          5: "B: own fate known", 6: "outputs issued"}),
    "gumbel_topk/two-launch keys": ([11, 12, 13, 14], {11: "keys start", 12: "loop done (loads, math, LDS hist)", 13: "hist added", 14: "end (statistics partial)"}),
    "gumbel_topk/two-launch emit": ([0, 1, 2, 3], {0: "emit start", 1: "selection returned", 2: "prefix recount done", 3: "outputs written"}),
    "frontier_expand_fused": ([0, 1, 2, 3, 4, 5], {0: "start", 1: "ids + count + row extents arrived (2 dependent trips)", 2: "row-length scan in LDS", 3: "side jobs issued (wg 0: eoff, marks, segments)",
                                                   4: "edges issued (column load -> stores, marks, in-degree atomics)", 5: "everything landed"}),
    "frontier_compact": ([0, 1, 7, 8, 2, 3, 4, 5, 6], {0: "start", 1: "bitmap words + word sums arrived", 7: "workgroup scan done", 8: "both look-back words published", 2: "counters requested (16 loads issued)", 3: "side jobs issued (marks, 3 MB of clears)",
                                              4: "predecessors' totals here (look-back)", 5: "emit issued", 6: "emit landed"}),
}


def show(name, idx, slots_key=None, ref_slot=None):
    tabs = records.get((name, idx))
    if not tabs:
        return
    slots, what = SLOTS[slots_key or name]
    st = np.stack(tabs[2:] if len(tabs) > 4 else tabs)
    ref = slots[0] if ref_slot is None else ref_slot
    live = st[:, :, ref] > 0
    if not live.any():
        return
    nb = int(live.sum(axis=1).max())
    t0 = np.where(live, st[:, :, ref], np.inf).min(axis=1)              # earliest start per batch
    print(f"{slots_key or name}  call {idx} of the step   ({nb} stamped workgroups, {st.shape[0]} batches)")
    last = None
    for sl in slots:
        ok = live & (st[:, :, sl] > 0)
        if not ok.any():
            print(f"   [{sl:2d}] {what[sl]:72s} (not on this path)")
            continue
        d = (st[:, :, sl] - t0[:, None]) * us
        dd = d[ok]
        lastwg = np.where(ok, d, -np.inf).max(axis=1)
        print(f"   [{sl:2d}] {what[sl]:72s} median {np.median(dd):6.2f} us   p90 {np.percentile(dd, 90):6.2f}   latest workgroup {np.median(lastwg):6.2f}")
        last = np.median(lastwg)
    print(f"        span, earliest start -> latest last stamp: {last:.2f} us")


print(f"# workload {pa.workload}; stamps in 100 MHz ticks -> us; eager launches of the real step, {pa.batches} batches (first two dropped)")
ncalls = max(i for (n_, i) in records if n_ == "gumbel_topk") + 1
for i in range(ncalls):
    show("gumbel_topk", i)
    show("gumbel_topk", i, "gumbel_topk/two-launch keys")
    show("gumbel_topk", i, "gumbel_topk/two-launch emit")
for i in range(max(i for (n_, i) in records if n_ == "frontier_expand_fused") + 1):
    show("frontier_expand_fused", i)
for i in range(max(i for (n_, i) in records if n_ == "frontier_compact") + 1):
    show("frontier_compact", i)

# per-workgroup view of the compaction's look-back (which workgroups are the stragglers): median over the batches of the stamps
# "published", "counters requested", "totals here", "emit issued" of every stamped workgroup of the hop-1 call
if pa.per_wg:
    tabs = records.get(("frontier_compact", 1))
    st = np.stack(tabs[2:] if len(tabs) > 4 else tabs)
    t0 = np.where(st[:, :, 0] > 0, st[:, :, 0], np.inf).min(axis=1)
    print("# frontier_compact call 1, per workgroup: published / requested / totals here / emit issued / emit landed (us, median over batches)")
    for b in range(st.shape[1]):
        if not (st[:, b, 0] > 0).any():
            continue
        row = [np.median((st[:, b, sl] - t0) * us) for sl in (8, 2, 4, 5, 6)]
        print(f"   wg {b:3d}  " + "  ".join(f"{v:6.2f}" for v in row))
