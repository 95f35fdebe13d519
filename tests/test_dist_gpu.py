"""The multi-GPU code path on ONE MI355X: a world_size-1 RCCL process group drives
dist.PartitionedGraph (all-to-all of adjacency rows + halo feature rows, gradient all-reduce) with
the real HIP kernels, and the whole training step must reproduce the single-GPU step bit for bit
on the index side and within 1e-5 on activations.  (Ranks > 1 are covered on the CPU over gloo by
tests/test_dist_cpu.py; the 2/4/8-GPU runs are the driver's.)"""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.fixture(scope="module")
def single_rank_group():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import torch.distributed as dist
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(_free_port())
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    yield dist
    if created:
        dist.destroy_process_group()


def test_partitioned_step_matches_single_gpu_step(single_rank_group):
    from grapes_amd import synth
    from grapes_amd.dist import make_grad_sync, shard_full_graph
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.step import GrapesTrainer
    n, deg, F, C, B, K, hops, H = 8000, 10.0, 24, 6, 64, 48, 2, 64
    indptr, indices = synth.synth_csr_numpy(n, deg, 400, seed=5)
    rng = np.random.default_rng(6)
    X = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.integers(0, C, n)).cuda()
    targets = torch.from_numpy(rng.permutation(n)[:B].astype(np.int64))
    uni = {h: torch.from_numpy(rng.random(n, dtype=np.float32)).cuda() for h in range(hops)}
    rowptr, col = torch.from_numpy(indptr).cuda(), torch.from_numpy(indices).cuda()

    def run(partitioned):
        torch.manual_seed(0)
        c, gf, z = GCN(F, [H, C]).cuda(), GCN(F + hops + 1, [H, 1]).cuda(), GCN(F, [H, 1]).cuda()
        if partitioned:
            g = shard_full_graph(rowptr, col, X, 0, 1, max_degree=int((rowptr[1:] - rowptr[:-1]).max()))
            tr = GrapesTrainer(g, None, y, c, gf, z, sampling_hops=hops, num_samples=K, loss_coef=10.0,
                               grad_sync=make_grad_sync(1))
        else:
            tr = GrapesTrainer(DeviceGraph(rowptr, col, n), X, y, c, gf, z, sampling_hops=hops, num_samples=K,
                               loss_coef=10.0)
        out = tr.step(targets, uniforms_fn=lambda h, nn: uni[h][:nn].contiguous(), trace=True)
        grads = [p.grad.clone() for m in (c, gf, z) for p in m.parameters()]
        return out, grads, tr

    a, ga, _ = run(False)
    b, gb, trb = run(True)
    for hop in range(hops):
        for key in ("neighborhoods", "batch_nodes", "neighbor_nodes", "local_neighborhoods", "kept", "k_hop_edges"):
            assert torch.equal(a["hops"][hop][key], b["hops"][hop][key]), (hop, key)
        assert torch.equal(a["hops"][hop]["indicator_rows"], b["hops"][hop]["indicator_rows"])
        assert torch.equal(a["hops"][hop]["cand_logits"], b["hops"][hop]["cand_logits"])     # halo rows are bit copies
    assert torch.equal(a["all_nodes"], b["all_nodes"])
    assert torch.equal(a["logits"], b["logits"])
    assert float(a["loss_c"]) == float(b["loss_c"]) and float(a["loss_gfn"]) == float(b["loss_gfn"])
    for x_, y_ in zip(ga, gb):
        assert torch.equal(x_, y_)
    assert trb.g.exchanged_bytes > 0


def test_explicit_step_over_partitioned_graph(single_rank_group):
    """The sync-free explicit-backward step (what bench.py runs for N > 1) over dist.PartitionedGraph — captured as
    hipGraph segments with the RCCL collectives between them — equals the same step run eagerly over the local
    DeviceGraph: sampled sets, losses and updated weights, through warm-up, capture and replays."""
    from grapes_amd import synth
    from grapes_amd.dist import make_grad_sync, shard_full_graph
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.step_graph import GraphedTrainer
    n, deg, F, C, B, K, hops, H = 12000, 10.0, 100, 6, 64, 48, 3, 128
    indptr, indices = synth.synth_csr_numpy(n, deg, 800, seed=9)
    rng = np.random.default_rng(10)
    X = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.integers(0, C, n)).cuda()
    batches = [torch.from_numpy(rng.permutation(n)[:B].astype(np.int64)).cuda() for _ in range(6)]
    rowptr, col = torch.from_numpy(indptr).cuda(), torch.from_numpy(indices).cuda()
    # (one of the batches holds a target WITHOUT any edge: it is one of all_nodes but a batch row of no hop — its features come
    # from the replicated rows of the isolated nodes, not from a hop's exchange: dist.PartitionedGraph.rows_from_kept)
    deg = np.diff(indptr)
    assert any(bool((deg[b.cpu().numpy()] == 0).any()) for b in batches)

    def run(partitioned):
        torch.manual_seed(0)
        c, gf, z = GCN(F, [H, H, C]).cuda(), GCN(F + hops + 1, [H, 1]).cuda(), GCN(F, [H, 1]).cuda()
        oc = torch.optim.Adam(c.parameters(), lr=1e-3, capturable=True)
        og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=1e-4, capturable=True)
        if partitioned:
            g = shard_full_graph(rowptr, col, X, 0, 1, max_degree=int((rowptr[1:] - rowptr[:-1]).max()))
            tr = GraphedTrainer(g, None, y, c, gf, z, batch_size=B, sampling_hops=hops, num_samples=K, loss_coef=20.0,
                                optimizer_c=oc, optimizer_gf=og, e_cap=1 << 14, philox_seed=5, grad_sync=make_grad_sync(1))
        else:
            tr = GraphedTrainer(DeviceGraph(rowptr, col, n), X, y, c, gf, z, batch_size=B, sampling_hops=hops,
                                num_samples=K, loss_coef=20.0, optimizer_c=oc, optimizer_gf=og, e_cap=1 << 14,
                                philox_seed=5, capture=False)
        outs = []
        for tg in batches:
            o = tr.step(tg)
            torch.cuda.synchronize()
            tr.check()
            outs.append(dict(kept=[k[:int(c_.item())].clone() for k, c_ in zip(o["kept"], o["kept_counts"])],
                             loss_c=float(o["loss_c"]), loss_gfn=float(o["loss_gfn"])))
        if partitioned:
            # per hop: row request / reply + feature request / reply; the last expansion's rows; the gradient all-reduce.  The
            # classifier's features are found among the hops' rows (dist.rows_from_kept): no fourth feature exchange
            assert tr.graph_obj is not None and tr.graph_obj.num_collectives == 4 * hops + 2 + 1
            assert tr.graph_obj.num_segments == tr.graph_obj.num_collectives + 1
            assert g.exchanged_bytes > 0
        return outs, [p.detach().clone() for m in (c, gf, z) for p in m.parameters()]

    a, wa = run(False)
    b, wb = run(True)
    for oa, ob in zip(a, b):
        for ka, kb in zip(oa["kept"], ob["kept"]):
            assert torch.equal(ka, kb)
        assert abs(oa["loss_c"] - ob["loss_c"]) <= 1e-5 * max(1.0, abs(oa["loss_c"]))
        assert abs(oa["loss_gfn"] - ob["loss_gfn"]) <= 1e-4 * max(1.0, abs(oa["loss_gfn"]))
    for p, q in zip(wa, wb):
        assert torch.allclose(p, q, rtol=1e-4, atol=1e-6)


def test_captured_step_over_peer_mapped_shards(single_rank_group):
    """X cut into 5 uneven row shards (each its own allocation, one empty) and read through peer.PeerFeatures' shard table by
    the fused gather-SpMM — the step bench.py runs for N > 1 with --halo peer, here with every shard inside one process — is the
    single-GPU captured step bit for bit (a row is the same 400 bytes wherever it lives): sampled sets, logits, losses,
    updated weights; ONE collective (the gradient all-reduce), two graph segments."""
    from grapes_amd import synth
    from grapes_amd.dist import make_grad_sync
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.peer import PeerFeatures
    from grapes_amd.step_graph import GraphedTrainer
    n, deg, F, C, B, K, hops, H = 12000, 10.0, 100, 6, 64, 48, 3, 128
    indptr, indices = synth.synth_csr_numpy(n, deg, 800, seed=9)
    rng = np.random.default_rng(10)
    X = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.integers(0, C, n)).cuda()
    batches = [torch.from_numpy(rng.permutation(n)[:B].astype(np.int64)).cuda() for _ in range(6)]
    rowptr, col = torch.from_numpy(indptr).cuda(), torch.from_numpy(indices).cuda()
    cuts = [0, 1000, 1000, 5000, 11000, n]

    def run(peers):
        torch.manual_seed(0)
        c, gf, z = GCN(F, [H, H, C]).cuda(), GCN(F + hops + 1, [H, 1]).cuda(), GCN(F, [H, 1]).cuda()
        oc = torch.optim.Adam(c.parameters(), lr=1e-3, capturable=True)
        og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=1e-4, capturable=True)
        Xa = PeerFeatures.from_shards([X[a:b].clone() for a, b in zip(cuts, cuts[1:])]) if peers else X
        tr = GraphedTrainer(DeviceGraph(rowptr, col, n), Xa, y, c, gf, z, batch_size=B, sampling_hops=hops, num_samples=K,
                            loss_coef=20.0, optimizer_c=oc, optimizer_gf=og, e_cap=1 << 14, philox_seed=5,
                            grad_sync=make_grad_sync(1))
        outs = []
        for tg in batches:
            o = tr.step(tg)
            torch.cuda.synchronize()
            tr.check()
            outs.append(dict(kept=[k[:int(c_.item())].clone() for k, c_ in zip(o["kept"], o["kept_counts"])],
                             logits=o["logits"][:int(o["n_all"])].clone(), loss_c=float(o["loss_c"]), loss_gfn=float(o["loss_gfn"])))
        assert tr.graph_obj is not None and tr.graph_obj.num_collectives == 1 and tr.graph_obj.num_segments == 2
        return outs, [p.detach().clone() for m in (c, gf, z) for p in m.parameters()]

    a, wa = run(False)
    b, wb = run(True)
    for oa, ob in zip(a, b):
        for ka, kb in zip(oa["kept"], ob["kept"]):
            assert torch.equal(ka, kb)
        assert torch.equal(oa["logits"], ob["logits"])
        assert oa["loss_c"] == ob["loss_c"] and oa["loss_gfn"] == ob["loss_gfn"]
    for p, q in zip(wa, wb):
        assert torch.equal(p, q)
    with pytest.raises(ValueError):          # transform-first first layers (F >= hidden) do not read through the table
        GraphedTrainer(DeviceGraph(rowptr, col, n), PeerFeatures.from_shards([X[:5000].clone(), X[5000:].clone()]), y,
                       GCN(F, [64, C]).cuda(), GCN(F + hops + 1, [64, 1]).cuda(), GCN(F, [64, 1]).cuda(), batch_size=B,
                       sampling_hops=hops, num_samples=K, capture=False)


def test_boundary_chained_steps_with_a_gradient_all_reduce(single_rank_group):
    """GraphedTrainer.run_steps on a step with ONE collective between its two graph segments (the N > 1 step of bench.py: peer-mapped
    shards + the RCCL gradient all-reduce, here at world size 1): the last segment of step t and the first segment of step t + 1 go out
    as one hipGraphLaunch, the all-reduce stays host-issued between them — 21 steps against 21 step_next() calls from the same state:
    weights, logits, sampled sets and edge totals EQUAL bit for bit; two boundary chains were built, no step chain."""
    from grapes_amd import synth
    from grapes_amd.dist import make_grad_sync
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.peer import PeerFeatures
    from grapes_amd.step_graph import GraphedTrainer
    n, deg, F, C, B, K, hops, H = 12000, 10.0, 100, 6, 64, 48, 3, 128
    indptr, indices = synth.synth_csr_numpy(n, deg, 800, seed=9)
    rng = np.random.default_rng(10)
    X = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.integers(0, C, n)).cuda()
    train = torch.from_numpy(rng.permutation(n)[:2000].astype(np.int64)).cuda()
    rowptr, col = torch.from_numpy(indptr).cuda(), torch.from_numpy(indices).cuda()
    cuts = [0, 1000, 5000, n]

    def run(chained, peers, host0=None):
        torch.manual_seed(0)
        c, gf, z = GCN(F, [H, H, C]).cuda(), GCN(F + hops + 1, [H, 1]).cuda(), GCN(F, [H, 1]).cuda()
        oc = torch.optim.Adam(c.parameters(), lr=1e-3, capturable=True)
        og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=1e-4, capturable=True)
        Xa = PeerFeatures.from_shards([X[a:b].clone() for a, b in zip(cuts, cuts[1:])]) if peers else X
        tr = GraphedTrainer(DeviceGraph(rowptr, col, n), Xa, y, c, gf, z, batch_size=B, sampling_hops=hops, num_samples=K,
                            loss_coef=20.0, optimizer_c=oc, optimizer_gf=og, e_cap=1 << 14, philox_seed=5,
                            grad_sync=make_grad_sync(1))
        tr.attach_loader(train)
        for _ in range(tr.eager_steps + 2):
            tr.step_next()
        assert tr._sets is not None and tr._sets[0].G is not None and tr._sets[0].G.num_collectives == 1
        if host0 is not None:                              # the epoch range runs out in the middle of the runs below
            for st in tr._sets:
                st.g._HOST0 = host0
            if chained:
                for k in (9, 7, 5):                        # (boundary-chained until a refill is near, plain steps across it, chained again)
                    out = tr.run_steps(k)
            else:
                for _ in range(21):
                    out = tr.step_next()
        elif chained:
            assert tr.prepare_chains(21) == 2
            out = tr.run_steps(20)
            out = tr.run_steps(1)                          # (a single step after a chained run: all three of its parts on their own)
            assert sorted(k for k in tr._chains) == [("b", 0), ("b", 1)]
        else:
            for _ in range(21):
                out = tr.step_next()
        torch.cuda.synchronize()
        tr.check()
        w = torch.cat([p.detach().view(-1) for m in (c, gf, z) for p in m.parameters()])
        return w, out["logits"].clone(), [k.clone() for k in out["kept"]], tr.edge_totals.clone(), tr.steps_done

    plain = {}
    for peers in (False, True):
        (wa, la, ka, ea, na), (wb, lb, kb, eb, nb) = run(True, peers), run(False, peers)
        plain[peers] = (wb, eb)
        assert na == nb and bool(torch.isfinite(wa).all()) and torch.equal(wa, wb), peers
        assert torch.equal(la, lb) and torch.equal(ea, eb), peers
        for p, q in zip(ka, kb):
            assert torch.equal(p, q)
    # ... and with the indicator table's epoch range running out inside the runs: the step that leaves the boundary-chained form
    # must not launch its first segment twice, the refill must not fall between a step's prelude and its main part
    (wc, lc, kc, ec, nc_), (wd, ld, kd, ed, nd) = run(True, False, host0=16), run(False, False, host0=16)
    assert nc_ == nd and torch.equal(wc, wd) and torch.equal(lc, ld) and torch.equal(ec, ed)
    assert torch.equal(wc, plain[False][0]) and torch.equal(ec, plain[False][1])     # (epochs are tags: the same 21 steps as above)
    for p, q in zip(kc, kd):
        assert torch.equal(p, q)


def test_two_process_partition_on_one_gpu():
    """TWO real processes on GPU 0 run the partitioned captured step with the HIP exchange kernels on both sides of every
    collective (gloo transport staged through the host: RCCL refuses two ranks on one device) — adjacency partitioned and
    adjacency replicated — and the peer-mapped form (each process reads the other's feature shard in place through a hipIpc
    mapping: no exchange, one collective) against the single-GPU step: tests/dist_gpu_worker.py."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "tests", "dist_gpu_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-5000:]
    for k in range(2):
        assert f"rank {k}/2 ok" in r.stdout, r.stdout[-3000:]
    assert "repl_adj ok" in r.stdout and "part_adj ok" in r.stdout and "peers ok" in r.stdout
