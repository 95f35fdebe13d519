#!/usr/bin/env python3
"""Soak: the pipelined captured step (prelude riders, backward carry, draws without a tail) against the one-graph captured step over
thousands of steps on the products-shaped workload — same seeds, same self-fed batches: the updated weights must be EQUAL bit for bit
at the end and every few hundred steps, the status word clean, the zero-at-rest tables zero.
usage: python profiles/soak_pipeline.py [steps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sys.argv = [sys.argv[0], "--cpu_steps", "0"]
import bench as B
args = B.parse()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
from grapes_amd import _lib, ops
_lib.load()
b = B.Bench(args, 1, 0, dev)


def make(pipeline):
    torch.manual_seed(0)
    tr, _, _ = b.make("single", seed=99, pipeline=pipeline)
    tr.attach_loader(b.train_idx, stride=1, offset=0)
    return tr

a, p = make(False), make(True)
assert p._pipeline_ok and not a._pipeline_ok
t0 = time.time()
for s in range(steps):
    a.step_next(); p.step_next()
    if (s + 1) % max(500, steps // 20) == 0 or s + 1 == steps:
        torch.cuda.synchronize()
        a.check(); p.check()
        same = all(torch.equal(x, y) for ma, mp in zip(a._models, p._models) for x, y in zip(ma.parameters(), mp.parameters()))
        fin = all(bool(torch.isfinite(x).all()) for m in p._models for x in m.parameters())
        print(f"step {s + 1}: weights equal {same}, finite {fin}, loss_c {float(p.out['loss_c']):.6f} / {float(a.out['loss_c']):.6f}, "
              f"pipelined graphs {p._sets is not None and p._sets[0].G is not None}", flush=True)
        assert same and fin
hist = ops._sampler_hist(dev)
tk = ops._ticket(dev)
print("histogram zero:", int(hist.ne(0).sum()) == 0, " tickets zero:", int(tk.ne(0).sum()) == 0, f" {time.time() - t0:.1f} s")
assert int(hist.ne(0).sum()) == 0 and int(tk.ne(0).sum()) == 0
print("soak ok")
