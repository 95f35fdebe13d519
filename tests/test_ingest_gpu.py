"""N3 (ingest): the device CSR builder against the SciPy-constructor semantics of main.py:134-136 (the oracle's
build_csr, itself pinned by golden G1), and the ogbn-papers100M SIZING run: N = 111,059,956 nodes — int64 row pointers, 14 MB
bitmaps, 444 MB id tables — with a reduced average degree so that it fits the test budget: the builder, one hop pipeline
(expand / compact / relabel / sample / slice) and a few captured training steps at that N."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import grapes_oracle as O

PAPERS_N = 111_059_956


def _cuda():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")


@pytest.mark.parametrize("N,E,seed", [(300, 4000, 0), (50_000, 400_000, 1), (7, 0, 2), (1_000_003, 3_000_000, 3)])
def test_csr_build_matches_scipy_constructor_semantics(N, E, seed):
    """Duplicates collapse, columns ascend, self-loops stay, empty rows, rows longer than one LDS window (hubs: sorted in
    place in global memory), a hub made of duplicates only."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(seed)
    parts = [rng.integers(0, N, (2, E))]
    if E:
        parts.append(parts[0][:, : E // 10])                                   # duplicates
        loops = rng.integers(0, N, E // 20); parts.append(np.stack([loops, loops]))   # self-loops
    if N >= 50_000:
        hub, hub2, hub3 = 17, N - 1, N // 2
        parts.append(np.stack([np.full(30_000, hub), rng.integers(0, N, 30_000)]))            # 30k entries, few duplicates
        parts.append(np.stack([np.full(9_000, hub2), rng.integers(0, 50, 9_000)]))            # 9k entries, 50 distinct
        parts.append(np.stack([np.full(4_097, hub3), np.arange(4_097) % N]))                  # just over one window
        parts.append(np.stack([np.arange(1000, 1400), np.full(400, 3)]))                      # a run of one-entry rows
    ei = np.concatenate(parts, axis=1).astype(np.int64)
    ei = ei[:, rng.permutation(ei.shape[1])]
    indptr, indices = O.build_csr(ei, N)
    rowptr, col = ops.csr_build(torch.from_numpy(ei).cuda(), N)
    assert rowptr.dtype == torch.int64 and col.dtype == torch.int32
    assert np.array_equal(rowptr.cpu().numpy(), indptr)
    assert np.array_equal(col.cpu().numpy(), indices)
    # ids out of range are reported, never written
    if E:
        bad = torch.from_numpy(ei).cuda().clone(); bad[1, 5] = N
        from grapes_amd import _lib
        with pytest.raises(_lib.GrapesHipError):
            ops.csr_build(bad, N)


def test_papers100m_sizing():
    """N = 111,059,956 (ogbn-papers100M), average degree ~4 after symmetrisation (the real graph: ~29): build the CSR on the
    device, check it against a device sort of the same keys, then run the hop pipeline and captured training steps on it."""
    _cuda()
    from grapes_amd import ops
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.step import GrapesTrainer
    from grapes_amd.step_graph import GraphedTrainer
    N, M = PAPERS_N, 220_000_000
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    # skewed endpoints (a few hubs with > 10^5 entries), symmetrised: 4.4e8 directed edges
    a = (torch.rand(M, device="cuda", generator=gen, dtype=torch.float64).pow_(3.0) * N).long().clamp_(0, N - 1)
    b = torch.randint(0, N, (M,), device="cuda", generator=gen)
    ei = torch.stack([torch.cat([a, b]), torch.cat([b, a])])
    del a, b
    rowptr, col = ops.csr_build(ei, N)
    nnz = int(rowptr[-1])
    assert rowptr.numel() == N + 1 and int(rowptr[0]) == 0 and col.numel() == nnz and nnz <= 2 * M
    deg = rowptr[1:] - rowptr[:-1]
    assert int(deg.min()) >= 0 and int(deg.max()) > 4096                      # hub rows went through the in-place path
    # every row strictly ascending: all adjacent pairs increase except across row boundaries
    inc = col[1:] > col[:-1]
    starts = rowptr[1:-1][deg[1:] > 0]                                         # first slot of every non-empty row but row 0
    inc[starts[starts < nnz] - 1] = True
    assert bool(inc.all())
    # same multiset of (row, col) as a device sort + unique of the 64-bit keys
    key = torch.unique(ei[0] * N + ei[1])
    assert key.numel() == nnz
    row_of = torch.div(key, N, rounding_mode="floor")
    assert torch.equal((key - row_of * N).to(torch.int32), col)
    assert torch.equal(torch.bincount(row_of[:50_000_000], minlength=1)[:1000], deg[:1000])
    del key, row_of, inc, ei
    torch.cuda.empty_cache()
    # ---- hop pipeline at this N (index properties, injected logits: test_products_scale_properties at papers100M size)
    dg = DeviceGraph(rowptr, col, N)
    B, K, hops, F = 256, 256, 3, 128                         # papers100M's real feature width: X is 56.9 GB
    targets = torch.randperm(N, device="cuda", generator=gen)[:B]
    from grapes_amd import synth
    X = synth.randn_rows_(torch.empty(N, F, device="cuda"), generator=gen)
    tr = GrapesTrainer(dg, X, None, None, None, None, sampling_hops=hops, num_samples=K)
    out = tr.step(targets, inject_logits_fn=lambda hop, bn: torch.sin(bn.to(torch.float32) * 0.001 + hop), trace=True)
    prev = targets
    for hop in range(hops):
        h = out["hops"][hop]
        nb = h["neighborhoods"]
        assert nb.shape[1] == int(deg[prev.long()].sum())
        bn = h["batch_nodes"].long()
        assert bool((bn[1:] > bn[:-1]).all()) and torch.equal(bn, torch.unique(nb.reshape(-1).long()))
        assert torch.equal(bn[h["local_neighborhoods"].long()], nb.long())
        kept = h["kept"].long()
        assert kept.numel() == min(K, h["neighbor_nodes"].numel()) and bool(torch.isin(kept, h["neighbor_nodes"].long()).all())
        prev = torch.cat([targets, kept])
    assert int(dg.bits.ne(0).sum()) == 0 and int(dg.mult.ne(0).sum()) == 0
    # ---- captured training steps (172 classes, 3-layer classifier: BASELINE config 5 on one GPU's worth of it)
    C, H = 172, 256
    y = torch.randint(0, C, (N,), device="cuda", generator=gen)
    torch.manual_seed(0)
    c, gf, z = GCN(F, [H, H, C]).cuda(), GCN(F + hops + 1, [H, 1]).cuda(), GCN(F, [H, 1]).cuda()
    oc = torch.optim.Adam(c.parameters(), lr=1e-3, capturable=True)
    og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=1e-4, capturable=True)
    gt = GraphedTrainer(dg, X, y, c, gf, z, batch_size=B, sampling_hops=hops, num_samples=K, loss_coef=100.0, optimizer_c=oc,
                        optimizer_gf=og, e_cap=1 << 20, philox_seed=3, capture=True)
    gt.attach_loader(torch.randperm(N, device="cuda", generator=gen)[:4096])
    losses = []
    for s in range(6):
        o = gt.step_next()
        torch.cuda.synchronize()
        gt.check()
        losses.append(float(o["loss_c"]))
        na = int(o["n_all"])
        alln = o["all_nodes"][:na].long()
        assert bool((alln[1:] > alln[:-1]).all()) and bool(torch.isin(gt.targets.long(), alln).all())
    assert gt.graph_obj is not None and all(np.isfinite(l) for l in losses)


def test_papers100m_full_edge_count_csr_and_steps(single_rank_group):
    """BASELINE config 5 at its REAL shape on one GPU: N = 111,059,956, average degree 29 symmetrised
    (~3.2e9 directed edges: column offsets beyond 2^31), F = 128, C = 172, 3 hops, GCN(128,[256,256,172]).  The CPU oracle
    cannot run at this size (a step is minutes, the arrays 70 GB), so the check is by size-independent properties: the
    device CSR (grapes_csr_build) has strictly ascending rows, no self-loops, is symmetric on sampled rows and keeps every
    distinct generated pair; the hop pipeline's expansions equal the CSR rows they query (offsets > 2^31 included); captured
    training steps run clean (status word, sorted all_nodes, finite losses).  `python bench.py --workload papers100m` times
    the same configuration (profiles/r03_bench_papers100m.json)."""
    _cuda()
    from grapes_amd import ops, synth
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.step import GrapesTrainer
    from grapes_amd.step_graph import GraphedTrainer
    N, deg, maxdeg, F, C, B, K, hops = synth.CONFIGS["papers100m"]
    assert N == PAPERS_N
    rowptr, col = synth.synth_graph_device_chunked(N, deg, maxdeg, seed=0, device="cuda")
    nnz = int(rowptr[-1])
    assert nnz == col.numel() and nnz > 3_000_000_000 and rowptr.dtype == torch.int64 and int(rowptr[0]) == 0
    d = rowptr[1:] - rowptr[:-1]
    assert int(d.min()) >= 0 and int(d.max()) > 4096                          # hub rows went through the in-place sort
    # every row strictly ascending, checked in slabs of 2^29 entries (a 3.2e9-element mask at once is 3 GB of bools)
    starts = rowptr[1:-1][d[1:] > 0]
    for lo in range(0, nnz - 1, 1 << 29):
        hi = min(nnz - 1, lo + (1 << 29))
        inc = col[lo + 1:hi + 1] > col[lo:hi]
        sb = starts[(starts > lo) & (starts <= hi)] - 1 - lo                   # pairs that straddle a row boundary
        inc[sb] = True
        assert bool(inc.all()), lo
        del inc, sb
    # sampled rows (some beyond the 2^31st entry): no self-loop, and symmetric — v in row(u)  =>  u in row(v)
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    far = torch.nonzero(rowptr[:-1] > (1 << 31))[:, 0]
    assert far.numel() > 0
    rows = torch.cat([torch.randint(0, N, (300,), device="cuda", generator=gen), far[torch.randint(0, far.numel(), (300,), device="cuda", generator=gen)]])
    rows = rows[d[rows] > 0]
    for u in rows[:200].tolist():
        nb = col[int(rowptr[u]):int(rowptr[u + 1])].long()
        assert bool((nb != u).all())
        for v in nb[:: max(1, nb.numel() // 4)].tolist():                       # a few neighbours per row
            rv = col[int(rowptr[v]):int(rowptr[v + 1])]
            j = int(torch.searchsorted(rv, torch.tensor([u], dtype=torch.int32, device="cuda")))
            assert j < rv.numel() and int(rv[j]) == u, (u, v)
    del starts, d
    torch.cuda.empty_cache()
    # ---- the hop pipeline over the full graph: expansions equal the queried CSR rows
    dg = DeviceGraph(rowptr, col, N)
    targets = torch.cat([far[torch.randint(0, far.numel(), (B // 2,), device="cuda", generator=gen)],
                         torch.randint(0, N, (B // 2,), device="cuda", generator=gen)]).unique()[:B]
    Xs = torch.zeros(N, 4, device="cuda")
    tr = GrapesTrainer(dg, Xs, None, None, None, None, sampling_hops=2, num_samples=K)
    out = tr.step(targets, inject_logits_fn=lambda hop, bn: torch.sin(bn.to(torch.float32) * 0.001 + hop), trace=True)
    prev = targets
    for hop in range(2):
        h = out["hops"][hop]
        nbh = h["neighborhoods"].long()
        want = torch.cat([col[int(rowptr[u]):int(rowptr[u + 1])].long() for u in prev.tolist()])
        assert torch.equal(nbh[1], want) and torch.equal(nbh[0], torch.repeat_interleave(prev.long(), (rowptr[prev.long() + 1] - rowptr[prev.long()])))
        bn = h["batch_nodes"].long()
        assert torch.equal(bn, torch.unique(nbh.reshape(-1)))
        prev = torch.cat([targets, h["kept"].long()])
    del Xs, tr, out
    torch.cuda.empty_cache()
    # ---- captured training steps at the real widths
    X = synth.randn_rows_(torch.empty(N, F, device="cuda"), generator=gen)
    y = torch.randint(0, C, (N,), device="cuda", generator=gen)
    H = 256
    torch.manual_seed(0)
    c, gf, z = GCN(F, [H, H, C]).cuda(), GCN(F + hops + 1, [H, 1]).cuda(), GCN(F, [H, 1]).cuda()
    oc = torch.optim.Adam(c.parameters(), lr=1e-3, capturable=True)
    og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=1e-4, capturable=True)
    gt = GraphedTrainer(dg, X, y, c, gf, z, batch_size=B, sampling_hops=hops, num_samples=K, loss_coef=100.0, optimizer_c=oc,
                        optimizer_gf=og, e_cap=1 << 18, philox_seed=3, capture=True)
    loader_ids = torch.randperm(N, device="cuda", generator=gen)[:8192]
    state0 = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in (c, gf, z)]
    gt.attach_loader(loader_ids)
    trace = []
    for s in range(8):
        o = gt.step_next()
        torch.cuda.synchronize()
        gt.check()
        na = int(o["n_all"])
        alln = o["all_nodes"][:na].long()
        assert bool((alln[1:] > alln[:-1]).all()) and bool(torch.isin(gt.targets.long(), alln).all())
        assert np.isfinite(float(o["loss_c"])) and np.isfinite(float(o["loss_gfn"]))
        for hop in range(hops):
            kc = int(o["kept_counts"][hop]); nn_ = int(o["sizes"][hop])
            kept = o["kept"][hop][:kc].long()
            assert kc == min(K, nn_) and bool(torch.isin(kept, o["neighbor_nodes"][hop][:nn_].long()).all())
        trace.append((alln.clone(), [o["kept"][hop][:int(o["kept_counts"][hop])].clone() for hop in range(hops)],
                      float(o["loss_c"]), float(o["loss_gfn"])))
    assert gt.graph_obj is not None
    assert torch.cuda.max_memory_allocated() < 200 * 2**30
    # ---- BASELINE config 5 in its PARTITIONED form: the same eight steps with X as EIGHT row shards read in
    # place through the peer table + the gradient all-reduce over a (world-1) RCCL group — what `bench.py --gpus 8 --workload
    # papers100m` runs per rank, minus the links.  Same weights, same loader, same Philox seed: the sampled sets and all_nodes must
    # be the single-GPU step's bit for bit, the losses equal to fp32 rounding of the all-reduced gradients' updates.
    from grapes_amd.dist import make_grad_sync, partition_bounds
    from grapes_amd.peer import PeerFeatures
    del gt, oc, og
    pb = partition_bounds(N, 8)
    shards = [X[a_:z_].clone() for a_, z_ in zip(pb, pb[1:])]
    del X
    torch.cuda.empty_cache()
    for m, st in zip((c, gf, z), state0):
        m.load_state_dict(st)
    oc = torch.optim.Adam(c.parameters(), lr=1e-3, capturable=True)
    og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=1e-4, capturable=True)
    dg = DeviceGraph(rowptr, col, N)           # (fresh per-graph state: the first trainer's last replay carried the prelude of a ninth step)
    gp = GraphedTrainer(dg, PeerFeatures.from_shards(shards), y, c, gf, z, batch_size=B, sampling_hops=hops, num_samples=K,
                        loss_coef=100.0, optimizer_c=oc, optimizer_gf=og, e_cap=1 << 18, philox_seed=3, capture=True,
                        grad_sync=make_grad_sync(1))
    gp.attach_loader(loader_ids)
    for s in range(8):
        o = gp.step_next()
        torch.cuda.synchronize()
        gp.check()
        alln, kept, lc, lg = trace[s]
        assert torch.equal(o["all_nodes"][:int(o["n_all"])].long(), alln), s
        for hop in range(hops):
            assert torch.equal(o["kept"][hop][:int(o["kept_counts"][hop])], kept[hop]), (s, hop)
        assert abs(float(o["loss_c"]) - lc) <= 1e-4 * max(1.0, abs(lc)) and abs(float(o["loss_gfn"]) - lg) <= 1e-3 * max(1.0, abs(lg)), (s, lc, lg)
    assert torch.cuda.max_memory_allocated() < 230 * 2**30
