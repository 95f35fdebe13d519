"""The BENCHED path under the oracle at the BASELINE configurations ("next round" item 1).

`GraphedTrainer(capture=True)` + `attach_loader` + `step_next()` — exactly what bench.py times: the self-feeding step
replayed as one hipGraph — against `oracle.grapes_oracle.train_step` (the CPU restatement of reference main.py:157-291)
for consecutive training iterations with both Adam optimisers, on synthetic graphs with the BASELINE.json shapes at FULL
size and the configs' real parameters:

  products  N=2,449,029  F=100  3 hops  K=256  B=256  classifier GCN(F,[256,256,47])   (north-star, BASELINE config 4)
  arxiv     N=169,343    F=128  2 hops  K=256  B=256  classifier GCN(128,[256,40])     (configs/gflownet/ogbn-arxiv.txt)
  reddit    N=232,965    F=602  2 hops  K=512  B=256  classifier GCN(602,[256,41])     (configs/gflownet/reddit.txt)
  cora      N=2,708      F=1433 2 hops  K=16   B=512  classifier GCN(1433,[256,7])     (code defaults, main.py:24-41)

The oracle draws its Gumbel uniforms from `portable_math.philox_uniform(seed, offset, n)` with the counter discipline of
the device sampler (offset += ceil(n/4) per draw that happens, i.e. when k < n), so both sides see identical bits.
Compared per step: the per-hop sampled sets (bit-exact), `all_nodes` (bit-exact), the sampler net's candidate logits and
the classifier logits (<= 1e-5 of the output scale on the first step), loss_c / log_z / sum log-prob / loss_gfn, every
gradient (<= 1e-4 of the largest entry) and the edges-aggregated count.  Steps 0-1 run eagerly (warm-up), step 2 is the
capture + first replay, step 3 a pure replay.

Later steps start from weights that went through two Adam implementations (torch's on the CPU, the fused launch here) fed
by gradients that agree to ~1e-6 relative; where |g| ~ eps Adam's update has slope lr*eps/(|g|+eps)^2, which moves a few
weights by ~1e-6 and activations by ~1e-5 — the value tolerances after step 0 are therefore 2e-4 (as in
test_captured_step_matches_eager_step); the index results stay exact.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import grapes_oracle as O
from oracle import portable_math as pm


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max())) if b.size else 0.0


_GATE_ALLOWANCE_USED = []          # (tensor name, units above tol, largest error) per comparison of the running test


def _margin(tag, what, got, ref, tol):
    """Records the headroom of one comparison: the norm-wise error max|a - b| / max(1, max|b|) the assertion
    uses, and the element-wise relative error over the entries with |b| > 1e-3 x the largest.  Printed (pytest -rP shows it) and,
    with GRAPES_PARITY_MARGINS_FILE set, appended to that file — profiles/r05_parity_margins.txt is one such run."""
    import os
    a = np.asarray(got, dtype=np.float64).reshape(-1); b = np.asarray(ref, dtype=np.float64).reshape(-1)
    if b.size == 0:
        return
    scale = max(1.0, float(np.abs(b).max()))
    nrm = float(np.abs(a - b).max()) / scale
    big = np.abs(b) > 1e-3 * float(np.abs(b).max())
    elem = float((np.abs(a - b)[big] / np.abs(b)[big]).max()) if big.any() else 0.0
    line = f"{tag:34s} {what:34s} normwise {nrm:9.2e} (tol {tol:7.1e}, {nrm / tol:6.3f} of it)   elementwise(|ref|>1e-3 max) {elem:9.2e}   n {b.size}"
    print(line)
    path = os.environ.get("GRAPES_PARITY_MARGINS_FILE")
    if path:
        with open(path, "a") as fh:
            fh.write(line + "\n")


def _grad_close(g, ref, tol, max_units=2, name=""):
    """Gradient comparison that knows about ReLU: d(loss)/d(pre-activation) is DISCONTINUOUS at 0, so a hidden unit m whose
    pre-activation in some row lies within fp32 rounding of 0 (a few 1e-7 of ~4e6 pre-activations per step: about one per
    step) may be gated differently by two correct fp32 implementations; that changes the gradient of exactly that unit's
    parameters (row m of W1, b1[m], w2[m]) by one row's contribution — observed: 3.6e-4 of the largest entry, confined to one
    row — and nothing else.  So: every unit within `tol` of the largest entry, except at most `max_units` units, which must
    stay within 10 x tol."""
    g = np.asarray(g, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    scale = max(1.0, float(np.abs(ref).max())) if ref.size else 1.0
    err = np.abs(g - ref) / scale
    if err.ndim == 2:                               # [units, in] (hidden layers) or [1, units] (the 1-wide heads)
        err = err.max(axis=1) if err.shape[0] > 1 else err.reshape(-1)
    bad = int((err > tol).sum())
    _GATE_ALLOWANCE_USED.append((name, bad, float(err.max(initial=0.0))))
    return bad <= max_units and float(err.max(initial=0.0)) <= 10 * tol, (bad, float(err.max(initial=0.0)))


def _captured_vs_oracle(workload, *, reinforce=False, use_indicators=True, multilabel=False, reg_param=0.0,
                        oracle_weights_each_step=False, steps=4, form="single"):
    """`GraphedTrainer(capture=True)` + loader against `O.train_step`, step for step.  `oracle_weights_each_step`: before
    every step after the first the ORACLE's post-update parameters are copied into the device modules, so that a step's
    kernels start from bit-identical weights whatever the two Adam implementations did to them — the first-step
    tolerances (1e-5 activations, 1e-4 gradients) must then hold at every step."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd import synth
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.step_graph import GraphedTrainer
    N, deg, maxdeg, F, C, B, K, hops = synth.CONFIGS[workload] if isinstance(workload, str) else workload      # (or a shape of its own)
    H, seed, coef = 256, 1234, 15227.124
    dev = torch.device("cuda")
    rowptr, col = synth.synth_graph_device(N, deg, maxdeg, seed=0, device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    X = torch.randn(N, F, device=dev, generator=gen)
    if multilabel:                                 # main.py:120-123: 2-D targets -> BCEWithLogitsLoss
        y = (torch.rand(N, C, device=dev, generator=gen) < 0.3).to(torch.float32)
    else:
        y = torch.randint(0, C, (N,), device=dev, generator=gen)
    n_train = max(4 * B, int(0.08 * N))
    train_idx = torch.randperm(N, device=dev, generator=gen)[:n_train]
    dims_c = [H] * (hops - 1) + [C]                # products: BASELINE's 3-layer classifier; 2 hops: main.py:110
    num_ind = hops + 1 if use_indicators else 0    # main.py:104-107,111-113
    torch.manual_seed(0)
    ref_c, ref_gf, ref_z = O.GCNRef(F, dims_c), O.GCNRef(F + num_ind, [H, 1]), O.GCNRef(F, [H, 1])
    c, gf, z = GCN(F, dims_c).to(dev), GCN(F + num_ind, [H, 1]).to(dev), GCN(F, [H, 1]).to(dev)
    c.load_state_dict(ref_c.state_dict()); gf.load_state_dict(ref_gf.state_dict()); z.load_state_dict(ref_z.state_dict())
    lr_c, lr_g = 4.469e-4, 2.556e-5                # configs/gflownet/ogbn-products.txt
    oc = torch.optim.Adam(c.parameters(), lr=lr_c, capturable=True)
    og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=lr_g, capturable=True)
    roc = torch.optim.Adam(ref_c.parameters(), lr=lr_c)
    rog = torch.optim.Adam(list(ref_gf.parameters()) + list(ref_z.parameters()), lr=lr_g)
    e_cap = 1 << 19 if workload == "reddit" else 1 << 17      # reddit: ~100 x 768 edges per hop + hubs
    # form: how the step sees the graph (BASELINE configs 4 / 5 name the PARTITIONED forms; no 8-GPU node exists here, so they run at
    # full size on one GPU: VERDICT r04 item 4).  "peer8": X as eight row shards read through the peer table (the N > 1 default) +
    # the gradient all-reduce over a world-1 RCCL group; "rccl": the request / reply halo exchange (--halo rccl) at world size 1.
    g_arg, X_arg, gs = DeviceGraph(rowptr, col, N), X, None
    if form == "peer8":
        from grapes_amd.dist import make_grad_sync, partition_bounds
        from grapes_amd.peer import PeerFeatures
        pb = partition_bounds(N, 8)
        X_arg, gs = PeerFeatures.from_shards([X[a_:z_].clone() for a_, z_ in zip(pb, pb[1:])]), make_grad_sync(1)
    elif form == "rccl":
        from grapes_amd.dist import make_grad_sync, shard_full_graph
        maxd = int((rowptr[1:] - rowptr[:-1]).max().item())
        g_arg, X_arg, gs = shard_full_graph(rowptr, col, X, 0, 1, max_degree=maxd, replicate_adjacency=True), None, make_grad_sync(1)
    tr = GraphedTrainer(g_arg, X_arg, y, c, gf, z, batch_size=B, sampling_hops=hops, num_samples=K,
                        loss_coef=coef, optimizer_c=oc, optimizer_gf=og, e_cap=e_cap, philox_seed=seed, capture=True,
                        reinforce_baseline=reinforce, use_indicators=use_indicators, reg_param=reg_param, grad_sync=gs)
    tr.attach_loader(train_idx)
    wl = workload if isinstance(workload, str) else "shape%s" % (tuple(workload),)
    variant = "+".join(v for v, on in (("reinforce", reinforce), ("noind", not use_indicators), ("multilabel", multilabel),
                                       ("reg", reg_param != 0.0), ("oracle-weights", oracle_weights_each_step), (form, form != "single")) if on)
    mtag = wl + ("/" + variant if variant else "")
    steps = max(steps, tr.eager_steps + 2)         # (a partitioned trainer warms up one step longer: the last steps must be replays)
    indptr, indices = rowptr.cpu().numpy(), col.cpu().numpy()
    Xc, yc, idx = X.cpu(), y.cpu(), train_idx.cpu().numpy()
    node_map = O.TensorMap(N)
    off = [0]

    def uniforms(hop, n):                           # the device sampler's counter discipline (sampler_kernels.hip)
        u = pm.philox_uniform(seed, off[0], n)
        off[0] += (n + 3) // 4
        return u

    for s in range(steps):
        if oracle_weights_each_step and s > 0:
            with torch.no_grad():
                for net, ref in ((c, ref_c), (gf, ref_gf), (z, ref_z)):
                    for p, q in zip(net.parameters(), ref.parameters()):
                        p.copy_(q.detach().to(dev))
            tr.weights_changed()                    # (the first layers' padded copies / split images follow the optimiser, not a copy_)
        out = tr.step_next()
        torch.cuda.synchronize()
        tr.check()
        tg = idx[(s * B) % max(1, n_train - B):][:B]
        assert np.array_equal(tr.targets.cpu().numpy().astype(np.int64), tg)
        ot = O.train_step(indptr, indices, Xc, yc, tg, ref_c, ref_gf, ref_z, sampling_hops=hops, num_samples=K,
                          uniforms_fn=uniforms, loss_coef=coef, optimizer_c=roc, optimizer_gf=rog, node_map=node_map,
                          reinforce_baseline=reinforce, use_indicators=use_indicators, reg_param=reg_param)
        tol = 1e-5 if (s == 0 or oracle_weights_each_step) else 2e-4
        for hop in range(hops):
            oh = ot["hops"][hop]
            kc = int(out["kept_counts"][hop])
            assert kc == len(oh["kept"]), (s, hop)
            assert np.array_equal(out["kept"][hop][:kc].cpu().numpy().astype(np.int64), oh["kept"]), (s, hop)   # bit-exact
            nn = int(out["sizes"][hop])
            assert nn == len(oh["neighbor_nodes"]), (s, hop)
            assert np.array_equal(out["neighbor_nodes"][hop][:nn].cpu().numpy().astype(np.int64), oh["neighbor_nodes"])
            cand = out["hop_logits"][hop].view(-1)[out["nb_local"][hop][:nn].long()].cpu().numpy()
            _margin(f"{mtag} step {s}", f"candidate logits hop {hop}", cand, oh["cand_logits"].numpy().reshape(-1), tol)
            assert _rel(cand, oh["cand_logits"].numpy().reshape(-1)) <= tol, (s, hop)             # layer activations
        na = int(out["n_all"])
        assert np.array_equal(out["all_nodes"][:na].cpu().numpy().astype(np.int64), ot["all_nodes"]), s
        _margin(f"{mtag} step {s}", "classifier logits", out["logits"][:na].cpu().numpy(), ot["logits"].numpy(), tol)
        assert _rel(out["logits"][:na].cpu().numpy(), ot["logits"].numpy()) <= tol, s
        for key, t in (("loss_c", tol), ("log_z", tol), ("tot_log_prob", 2 * tol), ("loss_gfn", 10 * tol)):
            assert abs(float(out[key]) - ot[key]) <= t * max(1.0, abs(ot[key])), (s, key, float(out[key]), ot[key])
        assert GraphedTrainer.edges_aggregated(out) == ot["edges_aggregated"], s
        gtol = 1e-4 if (s == 0 or oracle_weights_each_step) else 1e-3
        for name, net, ref in (("c", c, ref_c), ("gf", gf, ref_gf), ("z", z, ref_z)):
            if reinforce and name == "z":
                # main.py:277-279: the log-Z net takes no part in the REINFORCE loss: torch leaves its .grad at None (the
                # optimiser skips it), the captured step keeps zero gradients — and the weights must not move
                for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
                    assert q.grad is None or float(q.grad.abs().max()) == 0.0, (s, k)
                    assert float(p.grad.abs().max()) == 0.0, (s, k)
                continue
            for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
                _margin(f"{mtag} step {s}", f"grad {name}.{k}", p.grad.cpu().numpy(), q.grad.numpy(), gtol)
                if oracle_weights_each_step:        # first-step tolerance at every step, per hidden unit (see _grad_close)
                    ok, info = _grad_close(p.grad.cpu().numpy(), q.grad.numpy(), gtol, name=f"s{s}.{name}.{k}")
                    if info[0]:
                        print(f"{mtag} step {s}   gate allowance used by {name}.{k}: {info[0]} unit(s), largest {info[1]:.2e} (bound {10 * gtol:.1e})")
                    assert ok, (s, name, k, info)
                else:
                    assert _rel(p.grad.cpu().numpy(), q.grad.numpy()) <= gtol, (s, name, k)
    assert tr.graph_obj is not None                 # the last steps were graph replays
    return (c, ref_c, lr_c), (gf, ref_gf, lr_g), (z, ref_z, lr_g)


@pytest.mark.parametrize("workload", ["cora", "arxiv", "reddit", "products"])
def test_benched_captured_step_vs_oracle_at_baseline_configs(workload):
    nets = _captured_vs_oracle(workload)
    # Weights after four Adam updates on each side.  Adam's update lr*m/(sqrt(v)+eps) is ~ +-lr whatever |g| is, and has
    # slope lr*eps/(|g|+eps)^2 (up to lr/eps = 4e4) where |g| ~ eps = 1e-8: gradients that agree to 1e-9 absolute can move
    # such a weight by ~1e-5..1e-4 per step.  So: no weight may differ by more than a quarter of ONE update, and all but a
    # sliver must agree to fp32 accuracy.
    for net, ref, lr in nets:
        for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
            d = (p.detach().cpu() - q.detach()).abs()
            assert float(d.max()) <= 0.25 * lr, (k, float(d.max()))
            assert float((d > 1e-6 + 1e-5 * q.detach().abs()).float().mean()) <= 0.02, k


@pytest.mark.parametrize("form", ["peer8", "rccl"])
def test_partitioned_forms_of_the_products_step_vs_oracle_at_full_size(form, single_rank_group):
    """BASELINE config 4 in its PARTITIONED forms against the checker at FULL size on one GPU (no 8-GPU node
    exists for this build): the products step with X as EIGHT row shards read in place through the peer table + the gradient
    all-reduce over a (world-1) RCCL group — what `bench.py --gpus 8` runs per rank, minus the links — and with the RCCL request /
    reply halo exchange (`--halo rccl`: all-gather of id lists, all-to-all of rows, rows aggregated where they arrive).  Same
    assertions as the unpartitioned step: sampled sets and all_nodes bit-exact, activations 1e-5, gradients 1e-4, edge counts."""
    _captured_vs_oracle("products", form=form)


@pytest.mark.parametrize("shape", [(30000, 10.0, 500, 64, 5, 64, 32, 1), (30000, 10.0, 500, 48, 5, 32, 16, 4), (30000, 10.0, 500, 300, 5, 64, 32, 2)],
                         ids=["one-hop", "four-hops", "two-hops-transform-first"])
def test_captured_step_vs_oracle_at_other_hop_counts(shape):
    """--sampling_hops is a free parameter of the reference (main.py:110-114,178): one hop (no pipeline, one draw), four hops (the
    heads' backward at its capacity of four segments, a four-layer classifier) and a transform-first two-hop net against
    O.train_step, step for step — sampled sets bit-exact, activations 1e-5, gradients 1e-4 at the first step."""
    _captured_vs_oracle(shape, steps=4)


# The step variants SURVEY §8(f) N4 names, each against the ORACLE on the arxiv shape:
# REINFORCE loss (main.py:277-279), no indicator features (main.py:104-113,198-204), multilabel targets with
# BCEWithLogitsLoss (main.py:120-123), the logit-variance regulariser (main.py:260-261).
@pytest.mark.parametrize("variant", ["reinforce", "no_indicators", "multilabel", "reg_param", "reinforce_multilabel_reg"])
def test_benched_step_variants_vs_oracle(variant):
    kw = dict(reinforce=dict(reinforce=True), no_indicators=dict(use_indicators=False), multilabel=dict(multilabel=True),
              reg_param=dict(reg_param=0.05),
              reinforce_multilabel_reg=dict(reinforce=True, multilabel=True, reg_param=0.05))[variant]
    _captured_vs_oracle("arxiv", **kw)


@pytest.mark.parametrize("workload", ["arxiv", "products", "reddit", "cora"])
def test_kernels_hold_first_step_tolerances_at_later_steps_from_oracle_weights(workload):
    """the tolerances of the test above loosen after step 0 (1e-5 -> 2e-4, 1e-4 -> 1e-3) with an
    argument about Adam near |g| ~ eps.  Here both sides start EVERY step from the oracle's weights (copied to the device
    before the replay), which isolates the optimiser from the kernels: six steps, 1e-5 / 1e-4 throughout.  Reddit and Cora
     run the transform-first path — the gathered-operand bf16x3 GEMMs at K = 608 / 1436, the longest
    accumulation chains in the library — through the same steps (Reddit: four; its oracle step is the slowest).  The per-unit ReLU-gate allowance of `_grad_close` is
    measured: the number of units that used it is printed per run (pytest -s / -rP) and bounded over the whole run."""
    _GATE_ALLOWANCE_USED.clear()
    _captured_vs_oracle(workload, oracle_weights_each_step=True, steps=6 if workload != "reddit" else 4)
    used = sum(n for _, n, _ in _GATE_ALLOWANCE_USED)
    worst = max((e for _, _, e in _GATE_ALLOWANCE_USED), default=0.0)
    print(f"[gate allowance] {workload}: {used} unit(s) above 1e-4 in {len(_GATE_ALLOWANCE_USED)} tensor comparisons "
          f"(largest {worst:.2e}); per tensor: {[(k, n) for k, n, _ in _GATE_ALLOWANCE_USED if n]}")
    # ~one borderline pre-activation per step is expected (see _grad_close); a kernel error would show up in every tensor
    assert used <= 3 * 6, (used, _GATE_ALLOWANCE_USED)


@pytest.mark.parametrize("workload", ["arxiv", "products"])
def test_benched_random_sampling_step_vs_oracle_at_baseline_configs(workload):
    """The reference's configs/random/* mode (--random_sampling=True: uniform exact-k draws, no sampler / log-Z net,
    classifier update only; main.py:206-207,223,272) on the captured self-feeding step, same comparison as above."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd import synth
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.step_graph import GraphedTrainer
    N, deg, maxdeg, F, C, B, K, hops = synth.CONFIGS[workload]
    H, seed = 256, 77
    dev = torch.device("cuda")
    rowptr, col = synth.synth_graph_device(N, deg, maxdeg, seed=0, device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    X = torch.randn(N, F, device=dev, generator=gen)
    y = torch.randint(0, C, (N,), device=dev, generator=gen)
    n_train = max(4 * B, int(0.08 * N))
    train_idx = torch.randperm(N, device=dev, generator=gen)[:n_train]
    dims_c = [H] * (hops - 1) + [C]
    torch.manual_seed(0)
    ref_c = O.GCNRef(F, dims_c)
    c = GCN(F, dims_c).to(dev); c.load_state_dict(ref_c.state_dict())
    lr_c = 6.3169e-4                               # configs/random/ogbn-products.txt
    oc = torch.optim.Adam(c.parameters(), lr=lr_c, capturable=True)
    roc = torch.optim.Adam(ref_c.parameters(), lr=lr_c)
    tr = GraphedTrainer(DeviceGraph(rowptr, col, N), X, y, c, None, None, batch_size=B, sampling_hops=hops, num_samples=K,
                        optimizer_c=oc, e_cap=1 << 17, philox_seed=seed, capture=True, random_sampling=True)
    tr.attach_loader(train_idx)
    indptr, indices = rowptr.cpu().numpy(), col.cpu().numpy()
    Xc, yc, idx = X.cpu(), y.cpu(), train_idx.cpu().numpy()
    node_map = O.TensorMap(N)
    off = [0]

    def uniforms(hop, n):
        u = pm.philox_uniform(seed, off[0], n)
        off[0] += (n + 3) // 4
        return u

    for s in range(4):
        out = tr.step_next()
        torch.cuda.synchronize()
        tr.check()
        tg = idx[(s * B) % max(1, n_train - B):][:B]
        ot = O.train_step(indptr, indices, Xc, yc, tg, ref_c, None, None, sampling_hops=hops, num_samples=K,
                          uniforms_fn=uniforms, optimizer_c=roc, node_map=node_map, random_sampling=True)
        tol = 1e-5 if s == 0 else 2e-4
        for hop in range(hops):
            kc = int(out["kept_counts"][hop])
            assert np.array_equal(out["kept"][hop][:kc].cpu().numpy().astype(np.int64), ot["hops"][hop]["kept"]), (s, hop)
        na = int(out["n_all"])
        assert np.array_equal(out["all_nodes"][:na].cpu().numpy().astype(np.int64), ot["all_nodes"]), s
        assert _rel(out["logits"][:na].cpu().numpy(), ot["logits"].numpy()) <= tol, s
        assert abs(float(out["loss_c"]) - ot["loss_c"]) <= tol * max(1.0, abs(ot["loss_c"])), s
        assert out["loss_gfn"] is None and ot.get("loss_gfn") is None
        assert GraphedTrainer.edges_aggregated(out) == ot["edges_aggregated"], s
        for (k, p), (_, q) in zip(c.named_parameters(), ref_c.named_parameters()):
            assert _rel(p.grad.cpu().numpy(), q.grad.numpy()) <= (1e-4 if s == 0 else 1e-3), (s, k)
    assert tr.graph_obj is not None


def test_counted_build_peer_table_and_classic_build_give_the_same_captured_steps(monkeypatch):
    """Three forms of the captured step over one graph of 100k nodes (hubs, existing self-loops), six steps with both Adam
    optimisers each: (a) the hop graph from grapes_gcn_prepare's four launches (GRAPES_HOP_COUNTED=0), (b) the counted build
    (degree counting folded into the expansion and the compaction), (c) the counted build with X cut into 5 shards read through
    peer.PeerFeatures' table.  Sampled sets, logits, losses and the updated weights are EQUAL bit for bit: the builds write the
    same CSRs / head records and a row is the same bytes wherever it lives."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd import synth
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.peer import PeerFeatures
    from grapes_amd.step_graph import GraphedTrainer
    n, deg, F, C, B, K, hops, H = 100_000, 14.0, 100, 9, 128, 192, 3, 256
    indptr, indices = synth.synth_csr_numpy(n, deg, 6000, seed=3)
    # existing self-loops on every 11th node (replaced by the unit loop in gcn_norm)
    rows = np.repeat(np.arange(n), np.diff(indptr)); loops = np.arange(0, n, 11)
    ei = np.stack([np.concatenate([rows, loops]), np.concatenate([indices, loops])])
    indptr, indices = O.build_csr(ei, n)
    rng = np.random.default_rng(4)
    X = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.integers(0, C, n)).cuda()
    train = torch.from_numpy(rng.permutation(n)[:4000].astype(np.int64)).cuda()
    rowptr, col = torch.from_numpy(indptr).cuda(), torch.from_numpy(indices.astype(np.int32)).cuda()
    cuts = [0, 20_000, 20_000, 55_000, 90_001, n]

    def run(counted, peers):
        monkeypatch.setenv("GRAPES_DIAG", "1")      # (Python-side A/B switches are read only in a diagnostic session)
        monkeypatch.setenv("GRAPES_HOP_COUNTED", "1" if counted else "0")
        torch.manual_seed(0)
        c, gf, z = GCN(F, [H, H, C]).cuda(), GCN(F + hops + 1, [H, 1]).cuda(), GCN(F, [H, 1]).cuda()
        oc = torch.optim.Adam(c.parameters(), lr=1e-3, capturable=True)
        og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=1e-4, capturable=True)
        Xa = PeerFeatures.from_shards([X[a:b].clone() for a, b in zip(cuts, cuts[1:])]) if peers else X
        tr = GraphedTrainer(DeviceGraph(rowptr, col, n), Xa, y, c, gf, z, batch_size=B, sampling_hops=hops, num_samples=K,
                            loss_coef=30.0, optimizer_c=oc, optimizer_gf=og, e_cap=1 << 16, philox_seed=9)
        tr.attach_loader(train)
        outs = []
        for _ in range(6):
            o = tr.step_next()
            torch.cuda.synchronize()
            tr.check()
            outs.append(dict(kept=[k[:int(c_)].clone() for k, c_ in zip(o["kept"], o["kept_counts"])],
                             logits=o["logits"][:int(o["n_all"])].clone(), loss_c=float(o["loss_c"]), loss_gfn=float(o["loss_gfn"]),
                             edges=o["agg_counts"].clone()))
        assert tr.graph_obj is not None
        if counted:       # the counter tables are zero at rest
            hc = tr.g.hop_counters()
            for t in (hc.indeg, hc.loops, hc.wsum, hc.sync2):
                assert int(t.abs().max()) == 0
        return outs, [p.detach().clone() for m in (c, gf, z) for p in m.parameters()]

    a, wa = run(False, False)
    for counted, peers in ((True, False), (True, True)):
        b, wb = run(counted, peers)
        for s, (oa, ob) in enumerate(zip(a, b)):
            for ka, kb in zip(oa["kept"], ob["kept"]):
                assert torch.equal(ka, kb), (counted, peers, s)
            assert torch.equal(oa["logits"], ob["logits"]) and torch.equal(oa["edges"], ob["edges"]), (counted, peers, s)
            assert oa["loss_c"] == ob["loss_c"] and oa["loss_gfn"] == ob["loss_gfn"], (counted, peers, s)
        for p, q in zip(wa, wb):
            assert torch.equal(p, q), (counted, peers)


def _captured_trainer(workload, steps):
    from grapes_amd import synth
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.step_graph import GraphedTrainer
    N, deg, maxdeg, F, C, B, K, hops = synth.CONFIGS[workload]
    dev = torch.device("cuda")
    rowptr, col = synth.synth_graph_device(N, deg, maxdeg, seed=0, device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    X = torch.randn(N, F, device=dev, generator=gen)
    y = torch.randint(0, C, (N,), device=dev, generator=gen)
    train_idx = torch.randperm(N, device=dev, generator=gen)[:max(4 * B, int(0.08 * N))]
    torch.manual_seed(0)
    c, gf, z = GCN(F, [256] * (hops - 1) + [C]).to(dev), GCN(F + hops + 1, [256, 1]).to(dev), GCN(F, [256, 1]).to(dev)
    oc = torch.optim.Adam(c.parameters(), lr=4.469e-4, capturable=True)
    og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=2.556e-5, capturable=True)
    tr = GraphedTrainer(DeviceGraph(rowptr, col, N), X, y, c, gf, z, batch_size=B, sampling_hops=hops, num_samples=K,
                        loss_coef=15227.124, optimizer_c=oc, optimizer_gf=og,
                        e_cap=1 << 17 if workload in ("products", "arxiv", "cora") else 1 << 19, philox_seed=7, capture=True)
    tr.attach_loader(train_idx)
    for _ in range(steps):
        tr.step_next()
    torch.cuda.synchronize()
    tr.check()
    assert tr.graph_obj is not None
    return tr


@pytest.mark.parametrize("workload", ["reddit", "arxiv"])
def test_first_layer_copies_follow_the_weights_through_the_optimiser_launch(workload):
    """ops.FusedAdam(mirrors=...): the padded fp32 copy (arxiv: K = 131 -> 132) and the bf16x3 split image (Reddit: K = 605, 602) of
    a first layer's weight are written by the update launch itself — after captured steps they equal, BIT FOR BIT, what
    grapes_weight_split_image / a strided copy make of the updated weights, their padding untouched; and the step no longer holds
    the launches that refreshed them."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd import ops
    tr = _captured_trainer(workload, steps=5)
    assert tr._mirrors_current and tr._mirrored
    torch.cuda.synchronize()
    checked = 0
    for fl in tr._fl.values():
        w = fl.conv.lin.weight.detach()
        if id(fl.conv.lin.weight) not in tr._mirrored:
            continue
        if fl.padded:
            ref = torch.zeros_like(fl.W); ref[:, :fl.K] = w
            assert torch.equal(fl.W, ref)
            checked += 1
        if fl.split:
            pad = torch.zeros_like(fl.W) if fl.padded else None
            img = ops.weight_split_image(w, w_pad=pad)
            assert torch.equal(fl.image, img)
            checked += 1
    assert checked >= 1


@pytest.mark.parametrize("cfg", [dict(hops=1, F=100, H=256, B=128, K=64), dict(hops=4, F=100, H=256, B=64, K=32),
                                 dict(hops=2, F=300, H=256, B=128, K=64), dict(hops=2, F=100, H=256, B=128, K=64, ind=False),
                                 dict(hops=3, F=100, H=128, B=64, K=48, rnd=True), dict(hops=2, F=37, H=64, B=32, K=16)],
                         ids=["one-hop", "four-hops", "transform-first", "no-indicators", "random-sampling", "narrow"])
def test_pipelined_step_equals_one_graph_step_in_odd_configurations(cfg):
    """The self-feeding captured step with everything round 4 put inside other launches (the next step's prelude, the classifier's
    backward aggregations, the draws' tails, the optimiser's mirrors) against the same step as ONE graph without the prelude
    pipeline, away from the BASELINE shapes: one hop (no pipeline possible), four hops, a transform-first net (F >= hidden), no
    indicator columns, uniform draws, widths that are not multiples of 4 — after eight steps the weights of the three models are
    EQUAL bit for bit, finite, the status word clean."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd import synth
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.step_graph import GraphedTrainer
    dev = torch.device("cuda")
    hops, F, H, B, K = cfg["hops"], cfg["F"], cfg["H"], cfg["B"], cfg["K"]
    ind, rnd, N = cfg.get("ind", True), cfg.get("rnd", False), 60000

    def run(pipeline):
        rowptr, col = synth.synth_graph_device(N, 12.0, 2000, seed=0, device=dev)
        gen = torch.Generator(device=dev); gen.manual_seed(1)
        X = torch.randn(N, F, device=dev, generator=gen); y = torch.randint(0, 7, (N,), device=dev, generator=gen)
        train = torch.randperm(N, device=dev, generator=gen)[:4000]
        torch.manual_seed(0)
        ni = hops + 1 if ind else 0
        c, gf, z = GCN(F, [H] * (hops - 1) + [7]).to(dev), GCN(F + ni, [H, 1]).to(dev), GCN(F, [H, 1]).to(dev)
        oc = torch.optim.Adam(c.parameters(), lr=1e-3, capturable=True)
        og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=1e-4, capturable=True)
        tr = GraphedTrainer(DeviceGraph(rowptr, col, N), X, y, c, gf, z, batch_size=B, sampling_hops=hops, num_samples=K,
                            loss_coef=100.0, optimizer_c=oc, optimizer_gf=og, e_cap=1 << 16, philox_seed=3, capture=True,
                            use_indicators=ind, random_sampling=rnd, pipeline=pipeline)
        tr.attach_loader(train)
        for _ in range(8):
            tr.step_next()
        torch.cuda.synchronize()
        tr.check()
        w = torch.cat([p.detach().view(-1) for m in (c, gf, z) for p in m.parameters()])
        return w, (tr._sets is not None and tr._sets[0].G is not None)

    (wa, piped), (wb, _) = run(True), run(False)
    assert bool(torch.isfinite(wa).all()) and torch.equal(wa, wb)
    assert piped == (hops >= 2 and not rnd)


@pytest.mark.parametrize("cfg", [dict(hops=3, F=100, H=256, B=128, K=64), dict(hops=2, F=300, H=256, B=128, K=64),
                                 dict(hops=1, F=100, H=256, B=128, K=64), dict(hops=3, F=100, H=128, B=64, K=48, rnd=True)],
                         ids=["pipelined-three-hops", "transform-first", "one-graph-one-hop", "random-sampling"])
def test_chained_steps_equal_single_launch_steps(cfg):
    """GraphedTrainer.run_steps (include/grapes_hip.h: step chains — `chain` captured steps as ONE hipGraphLaunch, copies of the
    step graphs' kernel nodes) against the same number of step_next() calls: 3 warm-up steps, then 23 steps (two chains of 8, one
    of 6 and a single step; the pipelined trainers alternate their two sets inside a chain) — the weights of the three models,
    the last step's logits and sampled sets and the running edge totals are EQUAL bit for bit, the status word clean; a chain
    holds the kernel nodes of its steps and nothing else; prepare_chains builds what run_steps then uses."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd import synth
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.step_graph import GraphedTrainer
    dev = torch.device("cuda")
    hops, F, H, B, K = cfg["hops"], cfg["F"], cfg["H"], cfg["B"], cfg["K"]
    rnd, N = cfg.get("rnd", False), 60000

    def run(chained):
        rowptr, col = synth.synth_graph_device(N, 12.0, 2000, seed=0, device=dev)
        gen = torch.Generator(device=dev); gen.manual_seed(1)
        X = torch.randn(N, F, device=dev, generator=gen); y = torch.randint(0, 7, (N,), device=dev, generator=gen)
        train = torch.randperm(N, device=dev, generator=gen)[:4000]
        torch.manual_seed(0)
        c, gf, z = GCN(F, [H] * (hops - 1) + [7]).to(dev), GCN(F + hops + 1, [H, 1]).to(dev), GCN(F, [H, 1]).to(dev)
        oc = torch.optim.Adam(c.parameters(), lr=1e-3, capturable=True)
        og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=1e-4, capturable=True)
        tr = GraphedTrainer(DeviceGraph(rowptr, col, N), X, y, c, gf, z, batch_size=B, sampling_hops=hops, num_samples=K,
                            loss_coef=100.0, optimizer_c=oc, optimizer_gf=og, e_cap=1 << 16, philox_seed=3, capture=True,
                            random_sampling=rnd)
        tr.attach_loader(train)
        assert tr.prepare_chains(23) == 0                 # nothing is captured yet: nothing to build
        for _ in range(3):
            tr.step_next()
        if chained:
            built = tr.prepare_chains(23)
            assert built == 2                             # 8 + 8 + 6 (+ 1 single): the chains of 8 and of 6
            out = tr.run_steps(23, chain=8)
            assert len(tr._chains) == 2                   # ... and run_steps built no other
            per_step = {n: ch.nodes // n for (_, n), ch in tr._chains.items()}
            assert len(set(per_step.values())) == 1 and all(ch.nodes % n == 0 for (_, n), ch in tr._chains.items())
        else:
            for _ in range(23):
                out = tr.step_next()
        assert tr.steps_done == 26
        torch.cuda.synchronize()
        tr.check()
        w = torch.cat([p.detach().view(-1) for m in (c, gf, z) for p in m.parameters()])
        kept = [k.clone() for k in out["kept"]] if "kept" in out else []
        return w, out["logits"].clone(), kept, tr.edge_totals.clone()

    (wa, la, ka, ea), (wb, lb, kb, eb) = run(True), run(False)
    assert bool(torch.isfinite(wa).all()) and torch.equal(wa, wb)
    assert torch.equal(la, lb) and torch.equal(ea, eb)
    for p, q in zip(ka, kb):
        assert torch.equal(p, q)


def test_epoch_refills_in_the_middle_of_chained_and_pipelined_runs():
    """the indicator table's epoch range running out in the MIDDLE of run_steps.  The device hands out epochs
    1 .. _HOST0 - 3 per scratch set before the table is cleared and the counter restarts; with _HOST0 lowered to 24 a run of 70
    steps crosses that point three times per set — inside chains (which must stop short of it), between pipelined single steps
    (whose NEXT step's prelude is already riding: the refill has to be enqueued in FRONT of the graph that carries it), and in the
    one-graph trainer.  Chained, single-launch pipelined and one-graph runs end with EQUAL weights, logits, sampled sets and edge
    totals (epochs are tags: when the refill happens must not matter), and every run has refilled."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd import synth
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.step_graph import GraphedTrainer
    dev = torch.device("cuda")
    hops, F, H, B, K, N = 3, 100, 128, 64, 48, 40000

    def run(mode):
        rowptr, col = synth.synth_graph_device(N, 12.0, 2000, seed=0, device=dev)
        gen = torch.Generator(device=dev); gen.manual_seed(1)
        X = torch.randn(N, F, device=dev, generator=gen); y = torch.randint(0, 7, (N,), device=dev, generator=gen)
        train = torch.randperm(N, device=dev, generator=gen)[:4000]
        torch.manual_seed(0)
        c, gf, z = GCN(F, [H] * (hops - 1) + [7]).to(dev), GCN(F + hops + 1, [H, 1]).to(dev), GCN(F, [H, 1]).to(dev)
        oc = torch.optim.Adam(c.parameters(), lr=1e-3, capturable=True)
        og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=1e-4, capturable=True)
        tr = GraphedTrainer(DeviceGraph(rowptr, col, N), X, y, c, gf, z, batch_size=B, sampling_hops=hops, num_samples=K,
                            loss_coef=100.0, optimizer_c=oc, optimizer_gf=og, e_cap=1 << 16, philox_seed=3, capture=True,
                            pipeline=(mode != "one-graph"))
        tr.attach_loader(train)
        for _ in range(3):
            tr.step_next()
        graphs = [st.g for st in tr._sets] if tr._sets is not None else [tr.g]
        refills = []
        for g in graphs:
            g._HOST0 = 24
            zero = g.ind_code.zero_

            def counted(zero=zero):                        # (the table is only ever cleared by a refill)
                refills.append(1)
                return zero()
            g.ind_code.zero_ = counted
        if mode == "chained":
            for k in (23, 9, 38):
                out = tr.run_steps(k, chain=8)
        else:
            for _ in range(70):
                out = tr.step_next()
        assert tr.steps_done == 73
        torch.cuda.synchronize()
        tr.check()
        w = torch.cat([p.detach().view(-1) for m in (c, gf, z) for p in m.parameters()])
        return w, out["logits"].clone(), [k.clone() for k in out["kept"]], tr.edge_totals.clone(), len(refills)

    ref = run("one-graph")
    assert ref[4] >= 3 and bool(torch.isfinite(ref[0]).all())
    for mode in ("pipelined", "chained"):
        w, lg, kept, tot, nref = run(mode)
        assert nref >= 2, (mode, nref)                     # (two scratch sets, 35 epochs each: at least one refill per set)
        assert torch.equal(w, ref[0]) and torch.equal(lg, ref[1]) and torch.equal(tot, ref[3]), mode
        for p_, q_ in zip(kept, ref[2]):
            assert torch.equal(p_, q_)


def test_embed_nodes_captured_and_eager_steps_vs_oracle():
    """--embed_nodes (main.py:89-100,116): data.x is an nn.Parameter of optimizer_c.  The captured self-feeding step and the
    eager drop-in step, on the arxiv-shaped graph with a 64-wide embedding table, against O.train_step with the SAME parameter
    in its optimiser: sampled sets bit-exact, logits 1e-5, every net's gradients, the embedding gradient the classifier loss
    leaves (the dense [N, 64] matrix: zero outside all_nodes, 1e-4 of its largest entry inside) and the embeddings after
    Adam's dense update (rows that never received a gradient must not move at all)."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd import synth
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.step import GrapesTrainer
    from grapes_amd.step_graph import GraphedTrainer
    N, deg, maxdeg, _, C, B, K, hops = synth.CONFIGS["arxiv"]
    D, H, seed, coef = 64, 256, 77, 6414.70642460407
    dev = torch.device("cuda")
    rowptr, col = synth.synth_graph_device(N, deg, maxdeg, seed=0, device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    y = torch.randint(0, C, (N,), device=dev, generator=gen)
    train_idx = torch.randperm(N, device=dev, generator=gen)[:max(4 * B, int(0.08 * N))]
    indptr, indices = rowptr.cpu().numpy(), col.cpu().numpy()
    idx, yc = train_idx.cpu().numpy(), y.cpu()
    lr_c, lr_g = 0.0028881609333779408, 0.00015793805566708893       # configs/gflownet/blogcat.txt

    for engine in ("graph", "eager"):
        torch.manual_seed(0)
        emb0 = torch.randn(N, D)
        Xr = torch.nn.Parameter(emb0.clone())
        Xd = torch.nn.Parameter(emb0.clone().to(dev))
        ref_c, ref_gf, ref_z = O.GCNRef(D, [H, C]), O.GCNRef(D + hops + 1, [H, 1]), O.GCNRef(D, [H, 1])
        c, gf, z = GCN(D, [H, C]).to(dev), GCN(D + hops + 1, [H, 1]).to(dev), GCN(D, [H, 1]).to(dev)
        c.load_state_dict(ref_c.state_dict()); gf.load_state_dict(ref_gf.state_dict()); z.load_state_dict(ref_z.state_dict())
        oc = torch.optim.Adam(list(c.parameters()) + [Xd], lr=lr_c, capturable=True)
        og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=lr_g, capturable=True)
        roc = torch.optim.Adam(list(ref_c.parameters()) + [Xr], lr=lr_c)
        rog = torch.optim.Adam(list(ref_gf.parameters()) + list(ref_z.parameters()), lr=lr_g)
        g = DeviceGraph(rowptr, col, N)
        if engine == "graph":
            tr = GraphedTrainer(g, Xd, y, c, gf, z, batch_size=B, sampling_hops=hops, num_samples=K, loss_coef=coef,
                                optimizer_c=oc, optimizer_gf=og, e_cap=1 << 17, philox_seed=seed, capture=True)
            tr.attach_loader(train_idx)
        else:
            tr = GrapesTrainer(g, Xd, y, c, gf, z, sampling_hops=hops, num_samples=K, loss_coef=coef, optimizer_c=oc,
                               optimizer_gf=og, philox_seed=seed)
        node_map = O.TensorMap(N)
        off = [0]

        def uniforms(hop, n):
            u = pm.philox_uniform(seed, off[0], n)
            off[0] += (n + 3) // 4
            return u
        for s in range(4):
            tg = idx[(s * B) % max(1, len(idx) - B):][:B]
            if engine == "graph":
                out = tr.step_next()
            else:
                out = tr.step(torch.from_numpy(tg), trace=True)
            torch.cuda.synchronize()
            ot = O.train_step(indptr, indices, Xr, yc, tg, ref_c, ref_gf, ref_z, sampling_hops=hops, num_samples=K,
                              uniforms_fn=uniforms, loss_coef=coef, optimizer_c=roc, optimizer_gf=rog, node_map=node_map)
            tol = 1e-5 if s == 0 else 2e-4
            for hop in range(hops):
                if engine == "graph":
                    kc = int(out["kept_counts"][hop]); kept = out["kept"][hop][:kc]
                else:
                    kept = out["hops"][hop]["kept"]
                assert np.array_equal(kept.cpu().numpy().astype(np.int64), ot["hops"][hop]["kept"]), (engine, s, hop)
            na = int(out["n_all"])
            assert np.array_equal(out["all_nodes"][:na].cpu().numpy().astype(np.int64), ot["all_nodes"]), (engine, s)
            assert _rel(out["logits"][:na].cpu().numpy(), ot["logits"].numpy()) <= tol, (engine, s)
            assert abs(float(out["loss_c"]) - ot["loss_c"]) <= tol * max(1.0, abs(ot["loss_c"])), (engine, s)
            # the embedding gradient of the classifier loss: dense, zero outside all_nodes
            gx, rx = Xd.grad.cpu().numpy(), ot["x_grad_c"].numpy()
            touched = np.zeros(N, dtype=bool); touched[ot["all_nodes"]] = True
            assert float(np.abs(gx[~touched]).max()) == 0.0 and float(np.abs(rx[~touched]).max()) == 0.0, (engine, s)
            assert _rel(gx, rx) <= (1e-4 if s == 0 else 1e-3), (engine, s, _rel(gx, rx))
            for name, net, ref in (("c", c, ref_c), ("gf", gf, ref_gf), ("z", z, ref_z)):
                for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
                    assert _rel(p.grad.cpu().numpy(), q.grad.numpy()) <= (1e-4 if s == 0 else 1e-3), (engine, s, name, k)
            # the embeddings after the dense Adam update
            d = (Xd.detach().cpu() - Xr.detach()).abs()
            assert float(d.max()) <= 0.25 * lr_c, (engine, s, float(d.max()))
            assert float((d > 1e-6 + 1e-5 * Xr.detach().abs()).float().mean()) <= 0.002, (engine, s)
        moved = (Xd.detach().cpu() - emb0).abs().amax(dim=1) > 0
        assert 0 < int(moved.sum()) < N                     # the rows the steps touched moved (momentum included), the rest did not
        assert torch.equal(moved, (Xr.detach() - emb0).abs().amax(dim=1) > 0)
        if engine == "graph":
            assert tr.graph_obj is not None
