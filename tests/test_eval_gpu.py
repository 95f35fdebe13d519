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
    acc, f1, pred = evaluate(c, gf, data, args, g, mask=mask, full_batch=True, return_predictions=True)
    oacc, of1, opred = O.evaluate(indptr, indices, X, y, nodes, rc, rgf, sampling_hops=hops, num_samples=64, full_batch=True)
    assert float((pred.cpu() != opred).float().mean()) <= 2e-3        # (argmax of logits that agree to 1e-5: near-ties may flip)
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
    acc, f1, pred = evaluate(c, gf, data, args, g, mask=mask, loader=loader, full_batch=False, return_predictions=True)
    oacc, of1, opred = O.evaluate(indptr, indices, X, y, nodes, rc, rgf, sampling_hops=hops, num_samples=K, batch_size=B,
                                  full_batch=False)
    # the PREDICTIONS, node by node (equal accuracies could hide two wrong predictions that cancel)
    assert pred.shape == opred.shape and torch.equal(pred.cpu(), opred)
    assert abs(acc - oacc) < 1e-6 and abs(f1 - of1) < 1e-6
    assert int(g.bits.ne(0).sum()) == 0 and int(g.mult.ne(0).sum()) == 0


@pytest.mark.parametrize("hops", [2, 3])
def test_captured_minibatch_eval_matches_oracle_and_the_eager_loop(hops):
    """eval.py:71-163 as ONE captured step per batch (GraphedTrainer(evaluate=True): the training step's index chain with greedy
    draws — eval.py:126-130 — and the SWAPPED slice — eval.py:140-142 —, the classifier's forward pass and the targets' argmax;
    no host read per hop).  Nine full batches through the captured step + a ragged tenth through the eager loop, against the
    oracle's predictions node by node, and against the eager loop over all ten; the graph scratch is back at rest, and a second
    evaluation after the weights moved reuses the captured step and sees the new weights."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd.eval import evaluate
    n, F, C, H, K, B = 20000, 48, 6, 64, 24, 64
    ind = True
    indptr, indices, X, y, rc, rgf, c, gf, g, rng = _setup(n, 9.0, F, C, H, hops, 17)
    nodes = np.sort(rng.permutation(n)[:9 * B + 23])
    mask = torch.zeros(n, dtype=torch.bool); mask[torch.from_numpy(nodes)] = True
    loader = [(torch.from_numpy(nodes[i:i + B]),) for i in range(0, len(nodes), B)]
    data = types.SimpleNamespace(x=X, y=y)
    args = types.SimpleNamespace(sampling_hops=hops, num_samples=K, use_indicators=ind)
    acc, f1, pred = evaluate(c, gf, data, args, g, mask=mask, loader=loader, full_batch=False, return_predictions=True)
    tr = next(iter(g._eval_trainers.values()))
    assert tr.graph_obj is not None and tr.steps_done == 9            # nine batches went through the captured step
    oacc, of1, opred = O.evaluate(indptr, indices, X, y, nodes, rc, rgf, sampling_hops=hops, num_samples=K, batch_size=B,
                                  full_batch=False, use_indicators=ind)
    assert pred.shape == opred.shape and torch.equal(pred.cpu(), opred)
    assert abs(acc - oacc) < 1e-6 and abs(f1 - of1) < 1e-6
    acc2, _, pred2 = evaluate(c, gf, data, args, g, mask=mask, loader=loader, full_batch=False, return_predictions=True, captured=False)
    assert torch.equal(pred, pred2) and acc2 == acc
    assert int(g.bits.ne(0).sum()) == 0 and int(g.mult.ne(0).sum()) == 0 and int(g.prev_bits.ne(0).sum()) == 0
    g.check_status("captured evaluation")
    # the nets train between two evaluations (main.py:320-340): same captured step, new weights
    with torch.no_grad():
        for net, ref in ((c, rc), (gf, rgf)):
            for p_, q_ in zip(net.parameters(), ref.parameters()):
                q_.add_(0.05 * torch.randn(q_.shape, generator=torch.Generator().manual_seed(q_.numel())))
                p_.copy_(q_.to(p_.device))
    _, _, pred3 = evaluate(c, gf, data, args, g, mask=mask, loader=loader, full_batch=False, return_predictions=True)
    assert next(iter(g._eval_trainers.values())) is tr and tr.steps_done == 18
    _, _, opred3 = O.evaluate(indptr, indices, X, y, nodes, rc, rgf, sampling_hops=hops, num_samples=K, batch_size=B,
                              full_batch=False, use_indicators=ind)
    assert torch.equal(pred3.cpu(), opred3) and not torch.equal(opred3, opred)


def test_multilabel_full_batch_eval_matches_oracle_metrics():
    """eval.py:57-70 — multi-label targets: F1 from TP / FP / FN of `logit > 0` against `y > 0.5`, returned as accuracy and f1
    (0 on a zero denominator) — grapes_amd.eval._metrics against the oracle's restatement: the prediction matrices may differ
    only where a logit is within 1e-5 of 0, the F1 by what those entries can move it, and the degenerate cases are exact."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd.eval import evaluate, _metrics
    n, F, C, H, hops = 4000, 64, 11, 128, 2
    indptr, indices, X, y, rc, rgf, c, gf, g, rng = _setup(n, 10.0, F, C, H, hops, 5)
    ym = torch.from_numpy((rng.random((n, C)) < 0.3).astype(np.float32))
    nodes = np.sort(rng.permutation(n)[:1200])
    mask = torch.zeros(n, dtype=torch.bool); mask[torch.from_numpy(nodes)] = True
    data = types.SimpleNamespace(x=X, y=ym)
    args = types.SimpleNamespace(sampling_hops=hops, num_samples=32, use_indicators=True)
    acc, f1, pred = evaluate(c, gf, data, args, g, mask=mask, full_batch=True, return_predictions=True)
    oacc, of1, opred = O.evaluate(indptr, indices, X, ym, nodes, rc, rgf, sampling_hops=hops, num_samples=32, full_batch=True)
    assert acc == f1 and oacc == of1 and 0.0 < of1 < 1.0
    flips = int((pred.cpu() != opred).sum())
    assert flips <= 3, flips                                  # logits agree to 1e-5: only entries at |logit| < 1e-5 may differ
    tp_fp_fn = float((opred | (ym[torch.from_numpy(nodes)] > 0.5)).sum())
    assert abs(f1 - of1) <= 4.0 * (flips + 1e-9) / tp_fp_fn + 1e-12
    # the same logits through both implementations: exactly equal, including the zero-denominator branches (eval.py:69-70)
    lg = torch.from_numpy(rng.standard_normal((500, C)).astype(np.float32))
    yt = torch.from_numpy((rng.random((500, C)) < 0.4).astype(np.float32))
    for a, b in ((lg, yt), (-lg.abs() - 1, yt), (lg, torch.zeros_like(yt)), (lg.abs() + 1, torch.ones_like(yt))):
        assert _metrics(a.cuda(), b.cuda()) == O._metrics(a, b)


@pytest.mark.parametrize("B,C,n_rows", [(256, 47, 1022), (7, 172, 300), (64, 3, 64), (130, 65, 500)])
def test_eval_predict_equals_map_index_select_argmax(B, C, n_rows):
    """grapes_eval_predict (eval.py:154-155 in one launch) == node_map.map -> index_select -> torch.argmax, including rows with
    tied maxima (first index wins), NaNs (a NaN is the largest), -inf rows and widths that are not a multiple of the wavefront."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd import ops
    g = torch.Generator(device="cuda"); g.manual_seed(B * 1000 + C)
    logits = torch.randn(n_rows, C, device="cuda", generator=g)
    logits[3] = 0.5                                                   # all tied
    logits[5, C - 1] = logits[5].max() ; logits[5, 0] = logits[5, C - 1]   # tie between the first and the last column
    logits[7, C // 2] = float("nan")
    if C > 2:
        logits[9, 1] = float("nan"); logits[9, C - 1] = float("nan")
    logits[11] = float("-inf")
    logits[13, C - 1] = float("inf")
    N = 5000
    node_map = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    rows = torch.randperm(n_rows, device="cuda", generator=g)[:B].to(torch.int32)
    rows[: min(B, 7)] = torch.tensor([3, 5, 7, 9, 11, 13, 0], dtype=torch.int32, device="cuda")[: min(B, 7)]
    targets = torch.randperm(N, device="cuda", generator=g)[:B].to(torch.int32)
    node_map[targets.long()] = rows
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    pred, out = ops.eval_predict(logits, node_map, targets, status=status)
    ref_rows = logits.index_select(0, rows.long())
    assert torch.equal(torch.nan_to_num(out, nan=123.0), torch.nan_to_num(ref_rows, nan=123.0))
    assert torch.equal(pred, torch.argmax(ref_rows, dim=1)) and int(status) == 0
