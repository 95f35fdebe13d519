#!/usr/bin/env python3
"""The prelude pipeline (riders) against the plain one-graph step: same process, same box, alternating, completion time of
N step_next() calls per variant.  usage: pipeline_ab.py [workload]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wl = sys.argv[1] if len(sys.argv) > 1 else "products"
sys.argv = [sys.argv[0], "--cpu_steps", "0", "--workload", wl]
import bench as B
args = B.parse()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
from grapes_amd import _lib
_lib.load()
b = B.Bench(args, 1, 0, dev)
res = {True: [], False: []}
for rep in range(3):
    for pipe in (False, True):
        tr, g, models = b.make("single", seed=100, pipeline=pipe)
        tr.attach_loader(b.train_idx)
        for _ in range(800):
            tr.step_next()
        torch.cuda.synchronize(); tr.check()
        n = 1000
        t0 = time.perf_counter()
        for _ in range(n):
            tr.step_next()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        res[pipe].append(ms)
        print(f"pipeline={pipe}: {ms:.4f} ms/step  riders={[s.riders for s in tr._sets] if tr._sets else None}", flush=True)
        del tr
print("plain   :", " ".join(f"{v:.4f}" for v in res[False]))
print("pipeline:", " ".join(f"{v:.4f}" for v in res[True]))
