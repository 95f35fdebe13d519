#!/usr/bin/env python3
"""Where the eager drop-in loop (step.GrapesTrainer) spends its host time: cProfile over 40 steps of the products workload."""
import cProfile, os, pstats, sys, time
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
from grapes_amd.graph import DeviceGraph
from grapes_amd.step import GrapesTrainer
N, deg, maxdeg, F, C, Bs, K, hops = b.cfg
c, gf, z = B.build_models(F, 256, C, hops, dev)
oc = torch.optim.Adam(c.parameters(), lr=4.469e-4)
og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=2.556e-5)
tr = GrapesTrainer(DeviceGraph(b.rowptr, b.col, N), b.X, b.y, c, gf, z, sampling_hops=hops, num_samples=K, loss_coef=15227.124,
                   optimizer_c=oc, optimizer_gf=og, philox_seed=4321)
for s in range(10):
    tr.step(b.batch(s))
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for s in range(40):
    tr.step(b.batch(10 + s))
torch.cuda.synchronize()
pr.disable()
print("ms/step under cProfile: %.3f" % ((time.perf_counter() - t0) / 40 * 1e3))
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
