#!/usr/bin/env python3
"""Which C entry points does the PRODUCT call?  ("drop entry points the product build no longer calls".)
Runs, in one process with the product library and a counting proxy around it: bench.py's step on the five workloads (single GPU; the
peer-mapped and the RCCL-exchange forms of products at world size 1; the eager drop-in trainer; the reference-shaped loop), the CLI
(`grapes_amd.main`: GFlowNet, random sampling, REINFORCE, dropout, both engines, mini-batch and full-batch evaluation, embed_nodes),
evaluation, ingest.  Prints every entry point of the binding table with its call count; the ones at 0 are candidates for the
diagnostic build.   usage (GPU box): python profiles/entry_point_census.py > gpurun_out/entry_point_census.txt"""
import collections, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from grapes_amd import _lib

real = _lib.load()
counts = collections.Counter()


class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)
        if not name.startswith("grapes_"):
            return fn

        def wrapped(*a, **k):
            counts[name] += 1
            return fn(*a, **k)
        return wrapped


_lib._lib = Proxy()
sys.argv = [sys.argv[0]]
import bench
dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)


def section(name):
    print(f"# ... {name}: {sum(counts.values())} calls so far", file=sys.stderr, flush=True)


for wl in ("products", "arxiv", "reddit", "cora"):
    sys.argv = [sys.argv[0], "--workload", wl]
    args = bench.parse()
    b = bench.Bench(args, 1, 0, dev)
    modes = ["single"] + (["peer", "partition", "partition_adj"] if wl == "products" else [])
    for mode in modes:
        tr, g, _ = b.make(mode)
        tr.attach_loader(b.train_idx)
        for _ in range(tr.eager_steps + 3):
            tr.step_next()
        tr.run_steps(8, 4)
        torch.cuda.synchronize(); tr.check()
        del tr, g
        section(f"{wl}/{mode}")
    if wl in ("products", "cora"):
        tr, g, _ = b.make("single", capture=False, pipeline=False)
        tr.attach_loader(b.train_idx)
        tr.step_next(); tr.step_next()
        from grapes_amd.step import GrapesTrainer
        from grapes_amd.graph import DeviceGraph
        gc, gf, gz = bench.build_models(b.cfg[3], 256, b.cfg[4], b.cfg[7], dev)
        et = GrapesTrainer(DeviceGraph(b.rowptr, b.col, b.cfg[0]), b.X, b.y, gc, gf, gz, sampling_hops=b.cfg[7], num_samples=b.cfg[6])
        et.step(b.batch(0)); et.step(b.batch(1))
        section(f"{wl}/eager")
    del b
    torch.cuda.empty_cache()
from grapes_amd import main as cli
cli.main(["--dataset", "cora", "--max_epochs", "3", "--runs", "1", "--eval_frequency", "1", "--batch_size", "64", "--num_samples", "16",
          "--seed", "1", "--e_cap", "16384", "--hidden_dim", "64"])
for engine in ("graph", "eager"):
    cli.main(["--dataset", "cora", "--max_epochs", "1", "--runs", "1", "--batch_size", "64", "--num_samples", "8", "--random_sampling", "true",
              "--reg_param", "0.1", "--dropout", "0.2", "--seed", "2", "--max_steps", "3", "--hidden_dim", "32", "--eval_full_batch", "false",
              "--engine", engine])
cli.main(["--dataset", "cora", "--max_epochs", "1", "--runs", "1", "--batch_size", "64", "--num_samples", "8", "--reinforce_baseline", "true",
          "--dropout", "0.1", "--seed", "3", "--max_steps", "3", "--hidden_dim", "64", "--eval_full_batch", "false"])
cli.main(["--dataset", "cora", "--max_epochs", "1", "--runs", "1", "--batch_size", "64", "--num_samples", "8", "--embed_nodes", "true",
          "--node_emb_dim", "32", "--seed", "3", "--max_steps", "3", "--hidden_dim", "64"])
section("cli")
names = list(_lib.SIGNATURES)
print(f"# {len(names)} product entry points; calls counted over bench (5 workloads x forms), eager trainers, the CLI")
for n in sorted(names, key=lambda x: (counts[x] > 0, x)):
    print(f"{counts[n]:8d}  {n}")
print(f"# never called: {sum(counts[n] == 0 for n in names)}")
dist.destroy_process_group()
