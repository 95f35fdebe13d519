#!/usr/bin/env python3
"""Soak: GraphedTrainer.run_steps (32 captured steps per hipGraphLaunch: grapes_graph_chain_*) against one hipGraphLaunch per step
over tens of thousands of steps on the products-shaped workload — same seeds, same self-fed batches, both with the prelude pipeline:
the updated weights must be EQUAL bit for bit at every checkpoint, the edge totals equal, the status word clean, the zero-at-rest
tables zero.  Block lengths vary (chains of 32, 8, 2, odd remainders) so that chains and single steps are mixed.
usage: python profiles/soak_chain.py [steps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
sys.argv = [sys.argv[0], "--cpu_steps", "0"]
import bench as B
args = B.parse()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
from grapes_amd import _lib, ops
_lib.load()
b = B.Bench(args, 1, 0, dev)


def make():
    torch.manual_seed(0)
    tr, _, _ = b.make("single", seed=99, pipeline=True)
    tr.attach_loader(b.train_idx, stride=1, offset=0)
    return tr

a, c = make(), make()
blocks = [997, 1000, 33, 64, 1, 7, 512, 386]          # steps per round of the comparison (sum 3000)
t0, done, r = time.time(), 0, 0
while done < steps:
    k = min(blocks[r % len(blocks)], steps - done); r += 1
    for _ in range(k):
        a.step_next()
    c.run_steps(k, chain=(32, 8, 2)[r % 3])
    done += k
    if r % len(blocks) == 0 or done == steps:
        torch.cuda.synchronize()
        a.check(); c.check()
        same = all(torch.equal(x, y) for ma, mc in zip(a._models, c._models) for x, y in zip(ma.parameters(), mc.parameters()))
        fin = all(bool(torch.isfinite(x).all()) for m in c._models for x in m.parameters())
        tot = torch.equal(a.edge_totals, c.edge_totals)
        print(f"step {done}: weights equal {same}, edge totals equal {tot}, finite {fin}, loss_c {float(c.out['loss_c']):.6f} / "
              f"{float(a.out['loss_c']):.6f}, chains {sorted((k_, ch.nodes) for k_, ch in c._chains.items())}", flush=True)
        assert same and fin and tot and a.steps_done == c.steps_done
hist = ops._sampler_hist(dev)
tk = ops._ticket(dev)
print("histogram zero:", int(hist.ne(0).sum()) == 0, " tickets zero:", int(tk.ne(0).sum()) == 0, f" {time.time() - t0:.1f} s")
assert int(hist.ne(0).sum()) == 0 and int(tk.ne(0).sum()) == 0
print("soak ok")
