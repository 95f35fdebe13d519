#!/usr/bin/env python3
"""The prelude pipeline against the one-graph step, same process, same box: enqueue time (no sync) and completion time of N
step_next() calls per variant.  Variants: (pipeline, GRAPES_PIPE_MODE, GRAPES_PIPE_EVENT_FLAGS)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], "--cpu_steps", "0"]
import bench as B
args = B.parse()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
from grapes_amd import _lib
_lib.load()
b = B.Bench(args, 1, 0, dev)
variants = [(False, "overlap", "0"), (True, "overlap", "0"), (True, "overlap", hex(0x2 | 0x20000000)), (True, "overlap", "0x2"),
            (True, "serial", "0"), (False, "overlap", "0"), (True, "overlap", "0")]
for pipe, mode, flags in variants:
    os.environ["GRAPES_PIPE_MODE"] = mode; os.environ["GRAPES_PIPE_EVENT_FLAGS"] = flags
    tr, g, models = b.make("single", seed=100, pipeline=pipe)
    tr.attach_loader(b.train_idx)
    for _ in range(600):
        tr.step_next()
    torch.cuda.synchronize(); tr.check()
    n = 500
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step_next()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"pipeline={pipe} mode={mode} event_flags={flags}: enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, complete {1e3 * (t2 - t0) / n:.3f} ms/step", flush=True)
    del tr
