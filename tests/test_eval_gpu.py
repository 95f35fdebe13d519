"""Evaluation path (reference eval.py) on the MI355X against the CPU oracle: full-batch message passing over
the whole adjacency (the big gather-SpMM) and mini-batch greedy-sampler evaluation."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import grapes_oracle as O


def _setup(n, deg, F, C, H, hops, seed, asymmetric=False):
    from grapes_amd import synth
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    indptr, indices = synth.synth_csr_numpy(n, deg, 600, seed=seed)
    rng = np.random.default_rng(seed + 1)
    if asymmetric:   # drop a third of the entries: directed graph + add self-loops (PyG replaces them)
        rows = np.repeat(np.arange(n), np.diff(indptr))
        keep = rng.random(len(indices)) > 0.33
        ei = np.stack([np.concatenate([rows[keep], np.arange(0, n, 5)]), np.concatenate([indices[keep], np.arange(0, n, 5)])])
        indptr, indices = O.build_csr(ei, n)
    X = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, C, n))
    torch.manual_seed(seed)
    rc, rgf = O.GCNRef(F, [H] * (hops - 1) + [C]), O.GCNRef(F + hops + 1, [H, 1])
    c, gf = GCN(F, [H] * (hops - 1) + [C]).cuda(), GCN(F + hops + 1, [H, 1]).cuda()
    c.load_state_dict(rc.state_dict()); gf.load_state_dict(rgf.state_dict())
    c.eval(); gf.eval(); rc.eval(); rgf.eval()
    return indptr, indices, X, y, rc, rgf, c, gf, DeviceGraph.from_csr(indptr, indices), rng


@pytest.mark.parametrize("asymmetric", [False, True])
def test_full_batch_eval_matches_oracle(asymmetric):
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd.eval import evaluate
    n, F, C, H, hops = 6000, 100, 7, 256, 3
    indptr, indices, X, y, rc, rgf, c, gf, g, rng = _setup(n, 14.0, F, C, H, hops, 3, asymmetric)
    nodes = np.sort(rng.permutation(n)[:1500])
    mask = torch.zeros(n, dtype=torch.bool); mask[torch.from_numpy(nodes)] = True
    data = types.SimpleNamespace(x=X, y=y)
    args = types.SimpleNamespace(sampling_hops=hops, num_samples=64, use_indicators=True)
    acc, f1 = evaluate(c, gf, data, args, g, mask=mask, full_batch=True)
    oacc, of1, opred = O.evaluate(indptr, indices, X, y, nodes, rc, rgf, sampling_hops=hops, num_samples=64, full_batch=True)
    logits, _ = c(X.cuda(), g)                                                   # whole-graph logits, 1e-5
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
    rl, _ = rc(X, torch.from_numpy(np.stack([rows, indices.astype(np.int64)])))
    scale = max(1.0, float(rl.detach().abs().max()))
    assert float((logits.cpu() - rl.detach()).abs().max()) <= 1e-5 * scale
    with torch.inference_mode():           # the inference form (pre-scaled rows, line-padded pitch): same tolerance
        li, _ = c(X.cuda(), g)
    assert float((li.cpu() - rl.detach()).abs().max()) <= 1e-5 * scale
    assert abs(acc - oacc) < 1e-6 and abs(f1 - of1) < 1e-6
    assert g.gcn_prepared() is g.gcn_prepared()                                  # built once, cached


def test_minibatch_greedy_eval_matches_oracle():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd.eval import evaluate
    n, F, C, H, hops, K, B = 5000, 32, 5, 64, 2, 40, 128
    indptr, indices, X, y, rc, rgf, c, gf, g, rng = _setup(n, 9.0, F, C, H, hops, 7)
    nodes = np.sort(rng.permutation(n)[:300])
    mask = torch.zeros(n, dtype=torch.bool); mask[torch.from_numpy(nodes)] = True
    loader = [(torch.from_numpy(nodes[i:i + B]),) for i in range(0, len(nodes), B)]
    data = types.SimpleNamespace(x=X, y=y)
    args = types.SimpleNamespace(sampling_hops=hops, num_samples=K, use_indicators=True)
    acc, f1 = evaluate(c, gf, data, args, g, mask=mask, loader=loader, full_batch=False)
    oacc, of1, _ = O.evaluate(indptr, indices, X, y, nodes, rc, rgf, sampling_hops=hops, num_samples=K, batch_size=B,
                              full_batch=False)
    assert abs(acc - oacc) < 1e-6 and abs(f1 - of1) < 1e-6
    assert int(g.bits.ne(0).sum()) == 0 and int(g.mult.ne(0).sum()) == 0
