"""Command-line driver with the reference's flags and config files (reference main.py:23-54,367-390).

    python -m grapes_amd.main --config_file configs/gflownet/ogbn-products.txt --max_epochs 2

* Same flag names, defaults and explicit booleans (`--use_indicators true`) as the reference's `Arguments`
  (main.py:23-54); `--config_file` holds one `--flag value` pair per line (values may be quoted) and the command
  line takes precedence over it (main.py:370-374).
* Same experiment loop: `runs` x [ `max_epochs` epochs of sequential un-shuffled mini-batches over the training
  nodes (main.py:126,152-158), validation every `eval_frequency` epochs (main.py:320), a final test evaluation ],
  then `Acc: mean ± std` over the runs (main.py:376-390).
* The training iteration is grapes_amd's: GraphedTrainer (one captured hipGraph per step) for the GFlowNet sampler,
  its REINFORCE variant, `--random_sampling`, `--reg_param`, `--dropout` (masks from the sampler's Philox stream) and
  `--embed_nodes` (learned node embeddings in place of data.x, optimised by optimizer_c: main.py:89-100,116);
  `--engine eager` selects GrapesTrainer.

Datasets are outside this repository's scope (no dataset files and no network on the build machines): `--dataset`
names a SYNTHETIC graph with the statistics of the corresponding benchmark (grapes_amd.synth.CONFIGS — cora,
arxiv/ogbn-arxiv, reddit, products/ogbn-products), or a callable `module:function` returning an object with
`x, y, edge_index, train_mask, val_mask, test_mask` (the attributes main.py reads from its PyG `data`), so the
reference's own `data.get_data` can be plugged in where torch_geometric and the files exist.
"""
from __future__ import annotations

import argparse
import importlib
import shlex
import sys
import time
from types import SimpleNamespace
from typing import List, Optional, Sequence

import numpy as np
import torch

# (name, type, default) — main.py:23-54
_FLAGS = [
    ("dataset", str, "cora"), ("sampling_hops", int, 2), ("num_samples", int, 16), ("use_indicators", bool, True),
    ("lr_gf", float, 1e-4), ("lr_gc", float, 1e-3), ("loss_coef", float, 1e4), ("log_z_init", float, 0.0),
    ("reg_param", float, 0.0), ("dropout", float, 0.0), ("model_type", str, "gcn"), ("hidden_dim", int, 256),
    ("embed_nodes", bool, False), ("node_emb_dim", int, 64), ("max_epochs", int, 30), ("batch_size", int, 512),
    ("eval_frequency", int, 5), ("eval_on_cpu", bool, True), ("eval_full_batch", bool, True),
    ("random_sampling", bool, False), ("runs", int, 10), ("split_id", int, 0), ("seed", int, None),
    ("notes", str, None), ("log_wandb", bool, False), ("config_file", str, None), ("reinforce_baseline", bool, False),
]
# additions of this driver (not in the reference)
_EXTRA = [("e_cap", int, 1 << 17), ("max_steps", int, None), ("engine", str, "auto"), ("pipeline", bool, True)]

_DATASET_ALIASES = {"ogbn-arxiv": "arxiv", "ogbn-products": "products", "reddit2": "reddit"}


def _bool(v: str) -> bool:
    s = str(v).strip().lower()
    if s in ("true", "1", "yes"):
        return True
    if s in ("false", "0", "no"):
        return False
    raise argparse.ArgumentTypeError(f"expected true/false, got {v!r}")


def _parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="grapes_amd.main", description=__doc__.split("\n\n")[0])
    for name, typ, default in _FLAGS + _EXTRA:
        ap.add_argument(f"--{name}", type=_bool if typ is bool else typ, default=default)
    return ap


def read_config_file(path: str) -> List[str]:
    """`--flag value` per line, values optionally quoted (configs/gflownet/*.txt of the reference)."""
    out: List[str] = []
    with open(path) as f:
        for line in f:
            line = line.split("#", 1)[0].strip()
            if line:
                out.extend(shlex.split(line))
    return out


def parse_args(argv: Optional[Sequence[str]] = None) -> argparse.Namespace:
    """Reference semantics (main.py:367-374): if --config_file is given, the file's flags are read first and the
    command line is parsed again on top of them, so the command line wins."""
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = _parser()
    args = ap.parse_args(argv)
    if args.config_file is not None:
        args = ap.parse_args(read_config_file(args.config_file) + argv)
    if args.model_type != "gcn":
        raise NotImplementedError("only model_type=gcn is built (the reference's other model classes are dead code)")
    return args


# ------------------------------------------------------------------------------------------------ data
def synthetic_data(name: str, seed: int = 0, device="cuda") -> SimpleNamespace:
    """A graph with the named benchmark's statistics: features N(0,1), uniform labels, 10/5/85 % splits except
    products (8 % / 2 % / 90 %, the OGB proportions)."""
    from . import synth
    key = _DATASET_ALIASES.get(name.lower().strip('"'), name.lower().strip('"'))
    if key.startswith("synthetic:"):
        key = key.split(":", 1)[1]
    if key not in synth.CONFIGS:
        raise ValueError(f"unknown dataset {name!r}: synthetic stand-ins exist for {sorted(synth.CONFIGS)} "
                         "(or pass module:function)")
    N, deg, maxdeg, F, C, *_ = synth.CONFIGS[key]
    rowptr, col = synth.synth_graph_device(N, deg, maxdeg, seed=seed, device=device)
    gen = torch.Generator(device=device); gen.manual_seed(seed + 1)
    x = torch.randn(N, F, device=device, generator=gen)
    y = torch.randint(0, C, (N,), device=device, generator=gen)
    perm = torch.randperm(N, device=device, generator=gen)
    ftr, fva = (0.08, 0.02) if key == "products" else (0.10, 0.05)
    ntr, nva = int(ftr * N), int(fva * N)
    masks = [torch.zeros(N, dtype=torch.bool, device=device) for _ in range(3)]
    masks[0][perm[:ntr]] = True; masks[1][perm[ntr:ntr + nva]] = True; masks[2][perm[ntr + nva:]] = True
    return SimpleNamespace(x=x, y=y, rowptr=rowptr, col=col, edge_index=None, num_nodes=N, num_features=F,
                           num_classes=C, train_mask=masks[0], val_mask=masks[1], test_mask=masks[2], name=key)


def load_data(args, device) -> SimpleNamespace:
    if ":" in args.dataset and not args.dataset.lower().startswith("synthetic:"):
        mod, fn = args.dataset.split(":", 1)
        data = getattr(importlib.import_module(mod), fn)(args)
        if not hasattr(data, "num_classes"):
            data.num_classes = int(data.y.max().item()) + 1 if data.y.dim() == 1 else data.y.shape[1]
        data.num_features = data.x.shape[1]
        return data
    return synthetic_data(args.dataset, seed=args.seed or 0, device=device)


def _batches(idx: torch.Tensor, batch_size: int):
    for o in range(0, idx.numel(), batch_size):                   # DataLoader(TensorDataset(idx), batch_size) (main.py:126)
        yield idx[o:o + batch_size]


# ------------------------------------------------------------------------------------------------ one run
def train(args, device=None, log=print):
    from .eval import evaluate
    from .graph import DeviceGraph
    from .modules.gcn import GCN
    from .step import GrapesTrainer
    from .step_graph import GraphedTrainer
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    data = load_data(args, device)
    if getattr(data, "rowptr", None) is not None:
        g = DeviceGraph(data.rowptr, data.col, data.num_nodes)
    else:
        g = DeviceGraph.from_edge_index(data.edge_index.to(device), data.num_nodes)              # main.py:134-136
    y = data.y.to(device)
    if args.seed is not None:
        torch.manual_seed(args.seed)
    embedding_params = []
    if args.embed_nodes or getattr(data, "x", None) is None:                                       # main.py:89-100
        if not args.embed_nodes:
            raise ValueError("Dataset does not contain node features, and embed_nodes is False. "
                             "Did you mean to run with --embed_nodes=True?")                        # main.py:92-94
        log("Using learned node embeddings for features")
        emb = torch.empty(data.num_nodes, args.node_emb_dim)
        torch.nn.init.normal_(emb)                                                                 # main.py:96-97 (drawn on the host, like the reference)
        x = torch.nn.Parameter(emb.to(device), requires_grad=True)                                 # main.py:98-99: replaces data.x
        embedding_params.append(x)
    else:
        x = data.x.to(device).contiguous()
    F, C = x.shape[1], data.num_classes
    num_ind = args.sampling_hops + 1 if args.use_indicators else 0                                 # main.py:104-107
    gcn_c = GCN(F, hidden_dims=[args.hidden_dim, C], dropout=args.dropout).to(device)              # main.py:110
    gcn_gf = GCN(F + num_ind, hidden_dims=[args.hidden_dim, 1]).to(device)                          # main.py:112-113
    gcn_z = GCN(F, hidden_dims=[args.hidden_dim, 1]).to(device)                                     # main.py:114
    opt_c = torch.optim.Adam(list(gcn_c.parameters()) + embedding_params, lr=args.lr_gc, capturable=True)   # main.py:116
    opt_gf = torch.optim.Adam(list(gcn_gf.parameters()) + list(gcn_z.parameters()), lr=args.lr_gf, capturable=True)
    train_idx = data.train_mask.nonzero().squeeze(1)
    val_idx, test_idx = data.val_mask.nonzero().squeeze(1), data.test_mask.nonzero().squeeze(1)
    engine = args.engine
    if engine == "auto":
        engine = "graph"
    common = dict(sampling_hops=args.sampling_hops, num_samples=args.num_samples, use_indicators=args.use_indicators,
                  loss_coef=args.loss_coef, log_z_init=args.log_z_init, reinforce_baseline=args.reinforce_baseline,
                  optimizer_c=opt_c, optimizer_gf=opt_gf, philox_seed=args.seed or 0)
    # The captured step has a FIXED batch size.  A training split smaller than --batch_size (cora: 270 synthetic / 140
    # reference training nodes against the default 512) is one ragged batch, and every split ends in one (main.py:126 keeps
    # the DataLoader's last partial batch): the captured size is clamped to the split and ragged batches run through an
    # eager GrapesTrainer that shares models, optimisers (and their Adam state) and the graph with the captured one.
    batch_size = min(args.batch_size, int(train_idx.numel()))
    if batch_size <= 0:
        raise ValueError("the training split is empty")
    tail_trainer = None
    g_side = g          # the graph (scratch tables) of everything that runs BETWEEN captured steps: the ragged batch, evaluation
    if engine == "graph":
        trainer = GraphedTrainer(g, x, y, gcn_c, gcn_gf, gcn_z, batch_size=batch_size, e_cap=args.e_cap,
                                 random_sampling=args.random_sampling, reg_param=args.reg_param, pipeline=args.pipeline, **common)
        # the captured step feeds itself from the device-resident training ids — exactly the DataLoader's full batches, epoch
        # after epoch — and carries the next batch's weight-independent prelude: whatever else touches the graph's
        # bitmaps / relabel table in between gets scratch of its own (same CSR)
        trainer.attach_loader(train_idx, epochs=True)
        g_side = DeviceGraph(g.rowptr, g.col, g.num_nodes)
        g_side._max_degree = getattr(g, "_max_degree", None)
        if train_idx.numel() % batch_size:
            tail_trainer = GrapesTrainer(g_side, x, y, gcn_c, gcn_gf, gcn_z, random_sampling=args.random_sampling,
                                         reg_param=args.reg_param, **{**common, "philox_seed": (args.seed or 0) + 0x5eed})
    else:
        trainer = GrapesTrainer(g, x, y, gcn_c, gcn_gf, gcn_z, reg_param=args.reg_param,
                                random_sampling=args.random_sampling, **common)
    eval_args = SimpleNamespace(sampling_hops=args.sampling_hops, num_samples=args.num_samples,
                                use_indicators=args.use_indicators)
    edata = SimpleNamespace(x=x.detach(), y=y)          # (the learned embeddings are read in place: evaluation sees the updates)

    def run_eval(mask, idx):
        loader = [(b,) for b in _batches(idx, args.batch_size)]                                      # main.py:129,132
        return evaluate(gcn_c, gcn_gf, edata, eval_args, g_side, None, num_ind, device, mask, args.eval_on_cpu,
                        loader=loader, full_batch=args.eval_full_batch)

    steps = 0
    for epoch in range(1, args.max_epochs + 1):
        t0 = time.time()
        acc_c = torch.zeros((), device=device); acc_g = torch.zeros((), device=device)
        nb = 0
        for batch in _batches(train_idx, batch_size):
            if engine == "graph" and batch.numel() != batch_size:
                out = tail_trainer.step(batch)       # ragged last batch (main.py:126): same models / optimisers, eager
                if engine == "graph":
                    trainer.weights_changed()        # (the eager step's optimisers wrote the weights: the first layers' copies follow)
            elif engine == "graph":
                out = trainer.step_next()            # (the same batch: the loader walks train_idx in the DataLoader's order)
            else:
                out = trainer.step(batch)
            acc_c += out["loss_c"].reshape(()).detach()
            if out.get("loss_gfn") is not None:                      # absent under --random_sampling (main.py:206-207)
                acc_g += torch.as_tensor(out["loss_gfn"], device=device).reshape(()).detach()
            nb += 1; steps += 1
            if args.max_steps is not None and steps >= args.max_steps:
                break
        if engine == "graph":
            trainer.check()
        torch.cuda.synchronize()
        if nb == 0:
            raise RuntimeError("an epoch ran no training step")
        log(f"epoch {epoch}: loss_gfn={float(acc_g) / max(nb, 1):.6f}, loss_c={float(acc_c) / max(nb, 1):.6f}, "
            f"{nb} steps in {time.time() - t0:.2f}s")
        if (epoch + 1) % args.eval_frequency == 0:                                                   # main.py:320
            acc, f1 = run_eval(data.val_mask, val_idx)
            log(f"valid_accuracy={acc:.3f}, valid_f1={f1:.3f}")
        if args.max_steps is not None and steps >= args.max_steps:
            break
    acc, f1 = run_eval(data.test_mask, test_idx)                                                     # main.py:342-353
    log(f"test_accuracy={acc:.3f}, test_f1={f1:.3f}")
    return f1


def main(argv: Optional[Sequence[str]] = None) -> float:
    args = parse_args(argv)
    results = torch.empty(args.runs)
    for r in range(args.runs):                                                                       # main.py:376-385
        results[r] = train(args)
    std = float(results.std()) if args.runs > 1 else 0.0
    print(f"Acc: {100 * float(results.mean()):.2f} ± {100 * std:.2f}")                              # main.py:390
    return float(results.mean())


if __name__ == "__main__":
    main()
