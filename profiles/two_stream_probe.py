#!/usr/bin/env python3
"""How well do two INDEPENDENT captured steps overlap on two streams of one MI355X?  (Feasibility of running the next step's
weight-independent prefix under the current step's tail.)  Two products-shaped trainers with their own graph scratch, models and
optimisers: (a) both replayed on one stream, (b) one per stream, no dependency between them.  Prints ms per PAIR of steps."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B

# ROUND 4: HISTORICAL.  Two whole trainers share the per-device ticket / histogram scratch of ops.py: when their replays
# really co-run (default stream + a pool stream = two hardware queues) the kernels race and the run ends in a GPU memory
# fault (profiles/r04_overlap_probe.txt).  Kept for the record of round 3's measurement; refuses to run unless forced.
if os.environ.get("TWO_STREAM_FORCE", "0") != "1":
    raise SystemExit("two_stream_probe.py is a historical probe (see its header); the product's overlap is step_graph's prelude pipeline")
os.environ.setdefault("TWO_STREAM_DEFAULT", "0")
args = B.parse() if hasattr(B, "parse") else None
args.cpu_steps = 0
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
from grapes_amd import _lib, ops
_lib.load()
b = B.Bench(args, 1, 0, dev)
trs = []
for k in range(2):
    tr, g, models = b.make("single", seed=100 + k)
    tr.attach_loader(b.train_idx, stride=2, offset=k)
    if k == 1:      # its own look-back scratch: the two compactions may run at the same time
        ops.set_scratch_lane(1)
    for _ in range(6):
        tr.step_next()
    torch.cuda.synchronize(); tr.check()
    trs.append(tr)
# (round 4: two torch.cuda.Stream() objects land on the SAME hardware queue — rocprofv3 shows one queue id for both —, which is
# what round 3's "no overlap" measured; the default stream and one pool stream are different queues)
s0, s1 = (torch.cuda.default_stream(), torch.cuda.Stream()) if os.environ.get("TWO_STREAM_DEFAULT", "1") != "0" else (torch.cuda.Stream(), torch.cuda.Stream())

def run(two_streams, iters=300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        if two_streams:
            with torch.cuda.stream(s0): trs[0].step_next()
            with torch.cuda.stream(s1): trs[1].step_next()
        else:
            trs[0].step_next(); trs[1].step_next()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

for rep in range(2):
    print("one stream : %.3f ms per pair of steps" % run(False), flush=True)
    print("two streams: %.3f ms per pair of steps" % run(True), flush=True)
for tr in trs:
    tr.check()
print("status ok")
