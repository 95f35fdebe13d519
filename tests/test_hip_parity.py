"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle, the committed golden
fixtures and closed-form references.  Integer / index results must be bit-exact; floating-point
layer outputs are compared at the north-star tolerance of 1e-5 (relative to the output scale)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import grapes_oracle as O
from oracle import portable_math as pm

TOL = 1e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cuda():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    return torch.device("cuda")


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _t(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _close(a, b, tol=TOL):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
    return float(np.abs(a - b).max()) <= tol * scale if a.size else True


# ------------------------------------------------------------------------------ A4
def test_tensormap_golden(golden_dir):
    _cuda()
    from grapes_amd.modules.utils import TensorMap
    g = _load(golden_dir, "g3_tensormap.npz")
    tm = TensorMap(int(g["doc_keys"].max()) + 1)
    tm.update(torch.from_numpy(g["doc_keys"]))
    out = tm.map(torch.from_numpy(g["doc_query"]))
    assert out.dtype == torch.int64 and not out.is_cuda
    assert np.array_equal(out.numpy(), g["doc_out"])
    tm = TensorMap(64)
    tm.update(_t(g["seq_k1"]))
    assert np.array_equal(tm.map(_t(g["seq_k1"])).cpu().numpy(), g["seq_r1"])
    tm.update(_t(g["seq_k2"]))
    assert np.array_equal(tm.map(_t(g["seq_query"])).cpu().numpy(), g["seq_out"])   # stale entries persist


# ------------------------------------------------------------------------------ A1 / A3
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_get_neighborhoods_and_slice_golden(golden_dir, tag):
    _cuda()
    import scipy.sparse as sp
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.utils import get_neighborhoods, slice_adjacency
    g = _load(golden_dir, "g1_g2_csr.npz")
    n = int(g[f"{tag}_n"])
    # device CSR builder == SciPy constructor semantics (dedup + sort), main.py:134-136
    dg = DeviceGraph.from_edge_index(torch.from_numpy(g[f"{tag}_edge_index"]), n)
    assert np.array_equal(dg.rowptr.cpu().numpy(), g[f"{tag}_indptr"])
    assert np.array_equal(dg.col.cpu().numpy().astype(np.int64), g[f"{tag}_indices"])
    # the reference's own adjacency object is accepted as well
    A = sp.csr_matrix((np.ones(g[f"{tag}_indices"].shape[0], dtype=bool), g[f"{tag}_indices"], g[f"{tag}_indptr"]),
                      shape=(n, n))
    for adj in (dg, A):
        out = get_neighborhoods(torch.from_numpy(g[f"{tag}_nodes"]), adj)
        assert out.dtype == torch.int64
        assert np.array_equal(out.numpy(), g[f"{tag}_neigh"])
    out = get_neighborhoods(_t(g[f"{tag}_nodes_dup"]), dg)
    assert out.is_cuda and np.array_equal(out.cpu().numpy(), g[f"{tag}_neigh_dup"])
    rows, cols = torch.from_numpy(g[f"{tag}_rows"]), torch.from_numpy(g[f"{tag}_cols"])
    assert np.array_equal(slice_adjacency(dg, rows, cols).numpy(), g[f"{tag}_slice_rc"])
    assert np.array_equal(slice_adjacency(dg, cols, rows).numpy(), g[f"{tag}_slice_cr"])     # eval.py:140-142 order
    assert np.array_equal(slice_adjacency(dg, rows, torch.from_numpy(g[f"{tag}_cols_dup"])).numpy(), g[f"{tag}_slice_dup"])
    assert slice_adjacency(dg, rows, torch.zeros(0, dtype=torch.long)).shape == (2, 0)
    assert get_neighborhoods(torch.zeros(0, dtype=torch.long), dg).shape == (2, 0)
    assert int(dg.mult.abs().sum()) == 0      # scratch tables are clean at rest


def test_frontier_hub_and_ragged_rows():
    """Hub row (degree >> wavefront), empty rows, > 4096 query nodes (global-memory offset search)."""
    _cuda()
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.utils import get_neighborhoods
    rng = np.random.default_rng(3)
    n = 20000
    hub = np.stack([np.zeros(15000, np.int64), rng.permutation(n)[:15000]])
    rnd = rng.integers(0, n, (2, 60000))
    ei = np.concatenate([hub, rnd, rnd[::-1]], axis=1)
    indptr, indices = O.build_csr(ei, n)
    dg = DeviceGraph.from_csr(indptr, indices)
    for m in (1, 7, 513, 6000):
        nodes = rng.permutation(n)[:m].astype(np.int64)
        nodes[0] = 0
        ref = O.get_neighborhoods(nodes, indptr, indices)
        out = get_neighborhoods(_t(nodes), dg).cpu().numpy()
        assert np.array_equal(out, ref)


# ------------------------------------------------------------------------------ A8 compaction + gather
@pytest.mark.parametrize("n,m", [(500, 20), (70000, 700), (300000, 512)])
def test_frontier_compact_and_gather(n, m):
    dev = _cuda()
    from grapes_amd import ops
    from grapes_amd.graph import DeviceGraph
    rng = np.random.default_rng(n)
    ei = rng.integers(0, n, (2, n * 6))
    indptr, indices = O.build_csr(np.concatenate([ei, ei[::-1]], axis=1), n)
    dg = DeviceGraph.from_csr(indptr, indices)
    prev = rng.permutation(n)[:m].astype(np.int64)
    tm = O.TensorMap(n)
    nb, batch_nodes, neighbor_nodes, local = O.hop_index_pipeline(prev, indptr, indices, tm, n)
    p32 = _t(prev, torch.int32)
    eoff, d_e = ops.frontier_offsets(dg.rowptr, p32)
    e = int(d_e.item())
    assert e == nb.shape[1]
    src, dst, pos = ops.frontier_expand(dg.rowptr, dg.col, p32, eoff, e + 10, want_pos=True, status=dg.status)
    ops.bitmap_mark(dg.prev_bits, None, p32, n)
    ops.bitmap_mark(dg.bits, dg.bits1, src, n, d_n=d_e)
    ops.bitmap_mark(dg.bits, dg.bits1, dst, n, d_n=d_e)
    b, nbr, nbl, counts = ops.frontier_compact(dg.bits, dg.bits1, dg.prev_bits, n, e + m + 1, node_map=dg.node_map,
                                               status=dg.status)
    ops.bitmap_clear(dg.prev_bits, p32)
    c = counts.tolist()
    assert c == [len(batch_nodes), len(neighbor_nodes)]
    assert np.array_equal(b[:c[0]].cpu().numpy(), batch_nodes)                 # ascending global id (main.py:189)
    assert np.array_equal(nbr[:c[1]].cpu().numpy(), neighbor_nodes)            # main.py:190
    assert np.array_equal(nbl[:c[1]].cpu().numpy(), tm.map(neighbor_nodes))    # main.py:213
    assert np.array_equal(prev[pos[:e].cpu().numpy()], nb[0])
    lsrc = ops.tensormap_map(dg.node_map, src[:e].contiguous())
    ldst = ops.tensormap_map(dg.node_map, dst[:e].contiguous())
    assert np.array_equal(torch.stack([lsrc, ldst]).cpu().numpy(), local)      # main.py:195
    # bitmaps are consumed
    assert int(dg.bits.ne(0).sum()) == 0 and int(dg.prev_bits.ne(0).sum()) == 0
    assert int(dg.status.item()) == 0
    # feature gather with indicators (main.py:168,191,199-201)
    F, num_ind = 100, 3
    X = torch.randn(n, F, device=dev)
    ops.indicator_mark(dg.ind_code, p32[: m // 2].contiguous(), 5, num_ind - 1)
    ops.indicator_mark(dg.ind_code, nbr[:c[1]].contiguous(), 5, 0)
    ops.indicator_mark(dg.ind_code, p32[: m // 4].contiguous(), 4, 1)          # stale epoch: must read as 0
    ops.indicator_mark(dg.ind_code, p32[: m // 2].contiguous(), 5, num_ind - 1)
    x = ops.gather_rows(X, b[:c[0]].contiguous(), dg.ind_code, 5, num_ind)
    ind = np.zeros((n, num_ind), np.float32)
    ind[prev[: m // 2], -1] = 1
    ind[neighbor_nodes, 0] = 1
    ref = np.concatenate([X.cpu().numpy()[batch_nodes], ind[batch_nodes]], axis=1)
    assert np.array_equal(x.cpu().numpy(), ref)
    for Fo, ni in ((101, 2), (64, 0), (7, 1)):
        X2 = torch.randn(n, Fo, device=dev)
        x2 = ops.gather_rows(X2, b[:c[0]].contiguous(), dg.ind_code, 5, ni)
        ref = np.concatenate([X2.cpu().numpy()[batch_nodes], ind[batch_nodes][:, :ni]], axis=1)
        assert np.array_equal(x2.cpu().numpy(), ref)


def test_overflow_is_reported_not_silent():
    _cuda()
    from grapes_amd import ops, _lib
    from grapes_amd.graph import DeviceGraph
    rng = np.random.default_rng(0)
    n = 1000
    ei = rng.integers(0, n, (2, 20000))
    indptr, indices = O.build_csr(ei, n)
    dg = DeviceGraph.from_csr(indptr, indices)
    p32 = _t(np.arange(100), torch.int32)
    eoff, d_e = ops.frontier_offsets(dg.rowptr, p32)
    ops.frontier_expand(dg.rowptr, dg.col, p32, eoff, 50, status=dg.status)
    with pytest.raises(_lib.GrapesHipError):
        dg.check_status()


# ------------------------------------------------------------------------------ A2 sampler
def test_sampler_golden_bit_exact(golden_dir):
    _cuda()
    from grapes_amd import ops
    from grapes_amd.modules.utils import sample_neighborhoods_from_probs
    g = _load(golden_dir, "g4_sampler.npz")
    for tag in [str(s) for s in g["names"]]:
        logits, nodes, k = g[f"{tag}_logits"], g[f"{tag}_nodes"], int(g[f"{tag}_k"])
        n = nodes.shape[0]
        lt = _t(logits).requires_grad_(True)
        if k >= n:
            kept, lp, stats = sample_neighborhoods_from_probs(lt, torch.from_numpy(nodes), k)
            assert stats == {} and np.array_equal(kept.numpy(), g[f"{tag}_kept"])
            assert _close(lp.detach().cpu().numpy(), g[f"{tag}_logp"], 1e-6)
            continue
        r = _t(g[f"{tag}_uniforms"])
        kept, lp, stats = sample_neighborhoods_from_probs(lt, torch.from_numpy(nodes), k, uniforms=r)
        assert np.array_equal(kept.numpy(), g[f"{tag}_kept"]), tag                 # index set bit-exact, position order
        lpn, ref = lp.detach().cpu().numpy(), g[f"{tag}_logp"]
        assert np.array_equal(np.isinf(lpn), np.isinf(ref)), tag
        fin = np.isfinite(ref)
        assert np.allclose(lpn[fin], ref[fin], rtol=1e-5, atol=1e-6), tag
        st = np.array([float(stats[s]) for s in ("min_prob", "max_prob", "mean_entropy", "std_entropy")])
        assert np.allclose(st, g[f"{tag}_stats"], rtol=1e-4, atol=1e-6), tag
        # keys are bit-identical to the CPU oracle's portable arithmetic
        res = ops.gumbel_topk(_t(logits).reshape(-1), k, uniforms=r, want_keys=True)
        okeys = pm.gumbel_keys(logits.reshape(-1), g[f"{tag}_uniforms"])
        assert np.array_equal(res["keys"].cpu().numpy().view(np.uint32), okeys.view(np.uint32)), tag
        # backward of the Bernoulli log-prob: m - sigmoid(l)
        w = torch.randn(n, device="cuda")
        (lp * w).sum().backward()
        l64 = torch.from_numpy(logits.reshape(-1)).double()
        m = torch.from_numpy(res["mask"].cpu().numpy()).double()
        ref_g = (w.cpu().double() * (m - torch.sigmoid(l64))).numpy()
        assert _close(lt.grad.cpu().numpy().reshape(-1), ref_g, 1e-5), tag


def test_sampler_near_tie_family_device_equals_oracle_and_reference(golden_dir):
    """G6 (near ties down to one float32 ulp of the key, and exact ties): the device draw == the oracle's on every case, and
    == the reference's own kept set wherever the gap is > 0."""
    _cuda()
    from grapes_amd import ops
    g = _load(golden_dir, "g6_near_ties.npz")
    k, nodes = int(g["k"]), g["nodes"]
    ids = _t(nodes, torch.int32)
    for tag in g["names"]:
        logits, uni = _t(g[f"{tag}_logits"].reshape(-1)), _t(g[f"{tag}_uniforms"])
        res = ops.gumbel_topk(logits, k, uniforms=uni, candidate_ids=ids)
        kept = res["kept_ids"].cpu().numpy().astype(np.int64)
        o = O.sample_neighborhoods_from_probs(g[f"{tag}_logits"], nodes, k, g[f"{tag}_uniforms"])
        assert np.array_equal(kept, o["kept"]), str(tag)
        if float(g[f"{tag}_gap"]) > 0.0:
            assert np.array_equal(kept, g[f"{tag}_kept"]), str(tag)


@pytest.mark.parametrize("n,k,seed", [(70, 64, 0), (1025, 1, 1), (36543, 256, 2), (200000, 512, 3), (1 << 20, 256, 4)])
def test_sampler_vs_oracle_random(n, k, seed):
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(seed)
    logits = (rng.standard_normal(n) * 4).astype(np.float32)
    logits[rng.integers(0, n, max(1, n // 50))] = -150.0          # -inf keys
    logits[rng.integers(0, n, max(1, n // 100))] = 90.0
    r = rng.random(n, dtype=np.float32)
    ids = np.sort(rng.permutation(4 * n)[:n]).astype(np.int32)
    s = O.sample_neighborhoods_from_probs(logits, ids.astype(np.int64), k, r)
    prefix = (4 * n + np.arange(37)).astype(np.int32)
    res = ops.gumbel_topk(_t(logits), k, uniforms=_t(r), candidate_ids=_t(ids), want_keys=True, prefix_ids=_t(prefix))
    assert np.array_equal(res["union_ids"].cpu().numpy().astype(np.int64), np.concatenate([prefix, s["kept"]]))   # main.py:236-238
    assert int(res["union_count"].item()) == 37 + k
    assert abs(float(res["stats"][4]) - float(s["log_prob"].double().sum())) <= 1e-5 * max(1.0, abs(float(s["log_prob"].double().sum())))
    assert np.array_equal(res["keys"].cpu().numpy().view(np.uint32), s["keys"].view(np.uint32))
    assert np.array_equal(res["mask"].cpu().numpy() > 0.5, s["mask"])
    assert np.array_equal(res["kept_ids"].cpu().numpy().astype(np.int64), s["kept"])
    assert int(res["kept_count"].item()) == k
    kp = res["kept_pos"].cpu().numpy()
    assert np.all(np.diff(kp) > 0)                                              # candidate-position order
    assert _close(res["log_prob"].cpu().numpy(), s["log_prob"].numpy(), 1e-5)
    # size-independent property: every kept key >= every dropped key
    keys = res["keys"].cpu().numpy()
    o = pm.float_order_key(keys)
    assert o[s["mask"]].min() >= o[~s["mask"]].max()


@pytest.mark.parametrize("n,k,live", [(41000, 256, 40123), (3000, 256, 2999), (5000, 256, 200), (1500, 256, 1500), (70000, 64, 69999)])
def test_draw_without_a_tail_finished_by_the_expansion_that_follows(n, k, live):
    """grapes_gumbel_topk_deferred + grapes_frontier_expand_fused_finish: the draw's last launch has no ticket and no last
    workgroup; kept count, union count, stats[5] and the Philox advance come from its first workgroup, the log-prob sum (stats[4])
    and the histogram's return to zero from one extra workgroup of the expansion of the drawn nodes.  Everything — mask, kept ids in
    position order, the next query list, log-probs, all six statistics, the Philox counter, the expansion's own outputs — is equal
    BIT FOR BIT to the draw with a tail followed by the plain expansion; exact-k draws, keep-all draws (live count <= k), a live
    count far below the capacity, in-kernel Philox noise; the histogram and the tickets are zero afterwards."""
    _cuda()
    from grapes_amd import ops
    from grapes_amd.graph import DeviceGraph
    rng = np.random.default_rng(n + k)
    N = 50000
    ei = rng.integers(0, N, (2, N * 6))
    indptr, indices = O.build_csr(np.concatenate([ei, ei[::-1]], axis=1), N)
    dg = DeviceGraph.from_csr(indptr, indices)
    logits = _t((rng.standard_normal(n) * 3).astype(np.float32))
    ids = _t(np.sort(rng.permutation(N * 2)[:n] % N).astype(np.int32))
    prefix = _t(rng.permutation(N)[:100].astype(np.int32))
    d_n = torch.tensor([live], dtype=torch.int32, device="cuda")

    def run(defer):
        off = torch.tensor([12345], dtype=torch.int64, device="cuda")
        res = ops.gumbel_topk(logits, k, candidate_ids=ids, n=n, d_n=d_n, philox_seed=77, d_philox_offset=off, want_stats=True,
                              prefix_ids=prefix, defer_finish=defer)
        assert ("finish" in res) == defer
        rows, d_m = res["union_ids"], res["union_count"]
        src, dst, d_e, eoff = ops.frontier_expand_fused(dg.rowptr, dg.col, rows, 1 << 15, d_m=d_m, status=dg.status,
                                                        finish=res.get("finish"))
        torch.cuda.synchronize()
        kc, e = int(res["kept_count"]), int(d_e)
        return dict(mask=res["mask"][:live].clone(), kept=res["kept_ids"][:kc].clone(), kc=kc, uc=int(d_m), union=rows[:int(d_m)].clone(),
                    lp=res["log_prob"][:live].clone(), stats=res["stats"].clone(), off=int(off), e=e, src=src[:e].clone(), dst=dst[:e].clone(),
                    eoff=eoff[:int(d_m) + 1].clone())

    a, b = run(False), run(True)
    assert a["kc"] == b["kc"] == min(k, live) and a["uc"] == b["uc"] == 100 + min(k, live) and a["off"] == b["off"] and a["e"] == b["e"]
    for key in ("mask", "kept", "union", "lp", "src", "dst", "eoff"):
        assert torch.equal(a[key], b[key]), key
    assert torch.equal(a["stats"].view(torch.int32), b["stats"].view(torch.int32))      # bit for bit, the sum included
    assert float(a["stats"][5]) == (1.0 if live > k else 0.0)
    if live > k:
        assert a["off"] == 12345 + (live + 3) // 4
    assert int(ops._sampler_hist(torch.device("cuda", 0)).ne(0).sum()) == 0
    assert int(dg.status) == 0
    # the finish is the expansion's, whatever else rides in it: with the hop's bitmap marks and the previous-set clearing
    res = ops.gumbel_topk(logits, k, candidate_ids=ids, n=n, d_n=d_n, philox_seed=78, want_stats=True, prefix_ids=prefix, defer_finish=True)
    ops.frontier_expand_fused(dg.rowptr, dg.col, res["union_ids"], 1 << 15, d_m=res["union_count"], status=dg.status,
                              mark_prev_bits=dg.prev_bits, mark_bits=dg.bits, num_nodes=N, finish=res["finish"])
    ref = ops.gumbel_topk(logits, k, candidate_ids=ids, n=n, d_n=d_n, philox_seed=78, want_stats=True, prefix_ids=prefix)
    torch.cuda.synchronize()
    assert torch.equal(res["stats"].view(torch.int32), ref["stats"].view(torch.int32)) and torch.equal(res["mask"][:live], ref["mask"][:live])
    assert int(ops._sampler_hist(torch.device("cuda", 0)).ne(0).sum()) == 0


def test_sampler_ties_and_too_few_finite_keys():
    _cuda()
    from grapes_amd import ops
    n, k = 300, 64
    logits = np.full(n, -200.0, np.float32)          # all keys -inf except 10
    logits[np.arange(10) * 7 + 3] = 0.5
    r = np.random.default_rng(1).random(n, dtype=np.float32)
    s = O.sample_neighborhoods_from_probs(logits, np.arange(n), k, r)
    res = ops.gumbel_topk(_t(logits), k, uniforms=_t(r))
    assert np.array_equal(res["mask"].cpu().numpy() > 0.5, s["mask"])            # ties -> lowest positions
    assert int((res["mask"] > 0.5).sum()) == k
    # greedy (eval.py:126-130): top-k of probabilities
    logits = np.random.default_rng(2).standard_normal(5000).astype(np.float32)
    res = ops.gumbel_topk(_t(logits), 100, mode=1)
    ref = np.zeros(5000, bool)
    ref[np.argsort(-pm.p_sigmoid(logits), kind="stable")[:100]] = True
    assert np.array_equal(res["mask"].cpu().numpy() > 0.5, ref)


@pytest.mark.parametrize("case", ["clustered", "constant-keys", "greedy-saturated", "two-blocks", "all-in-last-block"])
def test_one_launch_draw_takes_the_scan_form_when_its_short_lists_cannot_hold_the_draw(case):
    """sampler_draw_k (round 5): a workgroup publishes at most 32 of its 512 keys — those at or above its own cut — and the
    draw's threshold comes from these lists.  Draws they cannot hold take the scan of the two-launch form INSIDE the same launch:
    one block of candidates owning most of the k largest keys (its cut lies above the selected bin), every key equal (one bin
    holds them all; ties go to the lowest positions), a greedy draw over saturated probabilities (eval.py:126-130: p == 1.0 a
    thousand times).  Against the oracle: masks, kept ids in position order, keys bit for bit."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(11)
    n, k, mode = 36000, 256, 0
    logits = (rng.standard_normal(n) * 2).astype(np.float32)
    if case == "clustered":
        logits[1536:2048] += 30.0                                # block 3 owns ~all of the 256 largest keys
    elif case == "constant-keys":
        logits[:] = 0.25
    elif case == "greedy-saturated":
        mode, k = 1, 300
        logits[rng.permutation(n)[:1000]] = 40.0                 # sigmoid == 1.0f: a thousand equal keys, k of them taken by position
    elif case == "two-blocks":
        n = 1000                                                 # two workgroups, the second one ragged
        logits = logits[:n]
    elif case == "all-in-last-block":
        logits[n - 200:] += 25.0; k = 128                        # the ragged last block holds every kept key
    r = rng.random(n, dtype=np.float32)
    if case == "constant-keys":
        r[:] = 0.5                                               # equal Gumbel noise too: all n keys are the SAME float
    ids = np.sort(rng.permutation(3 * n)[:n]).astype(np.int32)
    res = ops.gumbel_topk(_t(logits), k, uniforms=_t(r), candidate_ids=_t(ids), want_keys=True, mode=mode)
    mask = res["mask"].cpu().numpy() > 0.5
    assert int(mask.sum()) == k and int(res["kept_count"].item()) == k
    if mode == 1:
        ref = np.zeros(n, bool)
        ref[np.argsort(-pm.p_sigmoid(logits), kind="stable")[:k]] = True       # ties -> lowest positions
        assert np.array_equal(mask, ref)
    else:
        s = O.sample_neighborhoods_from_probs(logits, ids.astype(np.int64), k, r)
        assert np.array_equal(res["keys"].cpu().numpy().view(np.uint32), s["keys"].view(np.uint32))
        if case == "constant-keys":
            assert np.array_equal(np.nonzero(mask)[0], np.arange(k))            # every key equal: the first k positions
        else:
            assert np.array_equal(mask, s["mask"])
            assert _close(res["log_prob"].cpu().numpy(), s["log_prob"].numpy(), 1e-5)
    assert np.array_equal(res["kept_ids"].cpu().numpy(), ids[mask])            # candidate-position order
    assert np.array_equal(res["kept_pos"].cpu().numpy(), np.nonzero(mask)[0])
    assert int(ops._sampler_hist(torch.device("cuda", 0)).ne(0).sum()) == 0     # barrier words / ticket back at zero


def test_philox_matches_oracle():
    _cuda()
    from grapes_amd import ops
    for n, seed, off in ((1, 0, 0), (1000, 12345, 7), (100003, (1 << 40) + 5, (1 << 33) + 1)):
        a = ops.philox_uniform(n, seed, off, "cuda").cpu().numpy()
        b = pm.philox_uniform(seed, off, n)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert a.min() >= 0.0 and a.max() < 1.0
    # in-kernel generation == explicit uniforms
    n, k = 5000, 128
    logits = torch.randn(n, device="cuda")
    u = ops.philox_uniform(n, 99, 11, "cuda")
    r1 = ops.gumbel_topk(logits, k, uniforms=u)
    r2 = ops.gumbel_topk(logits, k, philox_seed=99, philox_offset=11)
    assert torch.equal(r1["mask"], r2["mask"])


# ------------------------------------------------------------------------------ A7 dense transforms
@pytest.mark.parametrize("n,fi,fo", [(1, 8, 8), (130, 104, 256), (1000, 100, 47), (257, 1433, 7), (4099, 256, 256),
                                     (777, 256, 1), (513, 103, 1), (300, 128, 130)])
def test_linear_fwd_bwd(n, fi, fo):
    _cuda()
    from grapes_amd import ops
    torch.manual_seed(n + fi)
    x = torch.randn(n, fi)
    w = torch.randn(fo, fi) / np.sqrt(fi)
    dh = torch.randn(n, fo)
    h = ops.linear_fwd(x.cuda(), w.cuda()).cpu()
    assert _close(h.numpy(), (x.double() @ w.double().t()).numpy())
    dw = ops.linear_bwd_weight(dh.cuda(), x.cuda()).cpu()
    ref_dw = (dh.double().t() @ x.double()).numpy()
    assert _close(dw.numpy(), ref_dw, 2e-5)
    dx = ops.linear_bwd_input(dh.cuda(), w.cuda()).cpu()
    assert _close(dx.numpy(), (dh.double() @ w.double()).numpy())
    # accumulate flag
    acc = torch.ones(fo, fi).cuda()
    ops.linear_bwd_weight(dh.cuda(), x.cuda(), out=acc, accumulate=True)
    assert _close(acc.cpu().numpy(), ref_dw + 1.0, 2e-5)
    # device-side row count: capacity-padded buffers, garbage beyond n must not matter
    cap = n + 77
    xp = torch.full((cap, fi), float("nan")); xp[:n] = x
    dhp = torch.full((cap, fo), float("nan")); dhp[:n] = dh
    d_n = torch.tensor([n], dtype=torch.int32).cuda()
    hp = ops.linear_fwd(xp.cuda(), w.cuda(), d_n=d_n).cpu()
    assert torch.equal(hp[:n], h)
    dwp = ops.linear_bwd_weight(dhp.cuda(), xp.cuda(), d_n=d_n).cpu()
    assert torch.equal(dwp, dw)


@pytest.mark.parametrize("n,fi,fo", [(1025, 256, 256), (1022, 256, 47), (1022, 100, 256), (129, 64, 96), (128, 13, 7), (4096, 104, 256), (5, 47, 256)])
def test_few_row_gated_dw_with_bias_sum(n, fi, fo):
    """dW = (dOut ⊙ [gate > 0])ᵀ X and db = its column sums for the classifier's few rows (one 128-row slab per workgroup,
    gate and bias sum inside the kernel), aligned and unaligned widths, a ragged last slab, accumulate, and a device-side row
    count over capacity-padded operands (NaN past the live rows must not matter); against fp64."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(n * 7 + fi + fo)
    x = torch.from_numpy(rng.standard_normal((n, fi)).astype(np.float32))
    dout = torch.from_numpy(rng.standard_normal((n, fo)).astype(np.float32))
    gate = torch.from_numpy(rng.standard_normal((n, fo)).astype(np.float32))
    A = dout.double() * (gate.double() > 0)
    ref_dw, ref_db = (A.t() @ x.double()).numpy(), A.sum(0).numpy()
    dw, db = ops.linear_bwd_weight_gated(dout.cuda(), x.cuda(), gate=gate.cuda())
    assert _close(dw.cpu().numpy(), ref_dw, 2e-5) and _close(db.cpu().numpy(), ref_db, 2e-5)
    dw1 = torch.full((fo, fi), 2.0, device="cuda"); db1 = torch.full((fo,), -1.0, device="cuda")
    ops.linear_bwd_weight_gated(dout.cuda(), x.cuda(), gate=gate.cuda(), dw=dw1, dbias=db1, accumulate=True)
    assert _close(dw1.cpu().numpy(), ref_dw + 2.0, 2e-5) and _close(db1.cpu().numpy(), ref_db - 1.0, 2e-5)
    cap = n + 131
    pad = lambda t: torch.cat([t, torch.full((cap - n, t.shape[1]), float("nan"))]).cuda()
    d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    dw2, db2 = ops.linear_bwd_weight_gated(pad(dout), pad(x), gate=pad(gate), d_n=d_n)
    assert _close(dw2.cpu().numpy(), ref_dw, 2e-5) and _close(db2.cpu().numpy(), ref_db, 2e-5)
    if cap <= 4096:          # same kernel (the dispatch goes by the CAPACITY): same slabs, same bits
        assert torch.equal(dw2, dw) and torch.equal(db2, db)
    dw3, _ = ops.linear_bwd_weight_gated(dout.cuda(), x.cuda(), want_bias=False)           # no gate, no bias sum
    assert _close(dw3.cpu().numpy(), (dout.double().t() @ x.double()).numpy(), 2e-5)


# ------------------------------------------------------------------------------ A6 / A7 GCNConv
def _rand_edges(rng, n, e, loops=True):
    ei = rng.integers(0, n, (2, e))
    if loops and n > 2:
        ei[:, : max(1, e // 20)] = rng.integers(0, n, max(1, e // 20))      # pre-existing self-loops
    return ei.astype(np.int64)


@pytest.mark.parametrize("n,e,fi,fo", [(9, 12, 5, 4), (600, 3000, 104, 256), (2000, 300, 100, 47), (1500, 20000, 256, 1),
                                       (3000, 9000, 64, 128), (100, 0, 16, 32)])
def test_gcn_conv_fwd_bwd_vs_oracle(n, e, fi, fo):
    _cuda()
    from grapes_amd.modules.gcn import GCNConv
    rng = np.random.default_rng(n * 7 + e)
    ei = _rand_edges(rng, n, e) if e else np.zeros((2, 0), np.int64)
    if n == 9:   # hand graph of the oracle's known-answer test: isolated row, hub, loop, duplicate
        ei = np.array([[0, 1, 2, 2, 3, 4, 5, 6, 7, 1, 1, 0], [1, 0, 2, 3, 1, 1, 1, 1, 1, 3, 3, 4]])
    if n == 1500:  # hub destination and hub source: rows >> GRAPES_LONG_ROW (workgroup-per-row path)
        ei[1, : e // 4] = 3
        ei[0, e // 4: e // 2] = 5
    torch.manual_seed(0)
    conv = GCNConv(fi, fo).cuda()
    with torch.no_grad():
        conv.bias.uniform_(-1, 1)
    x = torch.randn(n, fi)
    xg = x.cuda().requires_grad_(True)
    eig = torch.from_numpy(ei).cuda()
    for relu in (False, True):
        out = conv(xg, eig, relu=relu)
        W, b = conv.lin.weight.detach().cpu(), conv.bias.detach().cpu()
        ref64 = O.gcn_conv_dense_f64(x.numpy(), W.numpy(), b.numpy(), ei) if n <= 3000 else None
        xr = x.clone().requires_grad_(True)
        Wr, br = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
        ref = O.gcn_conv(xr, Wr, br, torch.from_numpy(ei))
        if relu:
            ref = torch.relu(ref)
            ref64 = np.maximum(ref64, 0)
        assert _close(out.detach().cpu().numpy(), ref.detach().numpy())
        assert _close(out.detach().cpu().numpy(), ref64)
        go = torch.randn(n, fo)
        conv.zero_grad(); xg.grad = None
        out.backward(go.cuda())
        ref.backward(go)
        assert _close(xg.grad.cpu().numpy(), xr.grad.numpy(), 2e-5)
        assert _close(conv.lin.weight.grad.cpu().numpy(), Wr.grad.numpy(), 2e-5)
        assert _close(conv.bias.grad.cpu().numpy(), br.grad.numpy(), 2e-5)
        # input without gradient (data features): the aggregate-first form act((ÂX)Wᵀ + b) with the fused
        # single-GEMM backward must give the same layer output and parameter gradients
        conv.zero_grad()
        out2 = conv(x.cuda(), eig, relu=relu)
        assert _close(out2.detach().cpu().numpy(), ref.detach().numpy())
        assert _close(out2.detach().cpu().numpy(), ref64)
        out2.backward(go.cuda())
        assert _close(conv.lin.weight.grad.cpu().numpy(), Wr.grad.numpy(), 2e-5)
        assert _close(conv.bias.grad.cpu().numpy(), br.grad.numpy(), 2e-5)


def test_gcn_prepare_structure():
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(5)
    n, e = 5000, 40000
    ei = _rand_edges(rng, n, e)
    ei[1, :3000] = 11        # long by-target row (wave-cooperative sort path)
    ei[0, 3000:5000] = 12    # long by-source row
    prep = ops.PreparedGraph(_t(ei[0], torch.int32), _t(ei[1], torch.int32), n)
    keep = ei[0] != ei[1]
    s, d = ei[0][keep], ei[1][keep]
    order = np.lexsort((s, d))
    rp = np.zeros(n + 1, np.int64); np.add.at(rp, d + 1, 1); rp = np.cumsum(rp)
    assert np.array_equal(prep.rowptr_t.cpu().numpy(), rp)
    assert np.array_equal(prep.csr_src.cpu().numpy()[: keep.sum()], s[order])     # ascending inside each row
    order = np.lexsort((d, s))
    rp = np.zeros(n + 1, np.int64); np.add.at(rp, s + 1, 1); rp = np.cumsum(rp)
    assert np.array_equal(prep.rowptr_s.cpu().numpy(), rp)
    assert np.array_equal(prep.csr_dst.cpu().numpy()[: keep.sum()], d[order])
    deg = np.bincount(d, minlength=n) + 1
    assert np.allclose(prep.dinv.cpu().numpy(), 1.0 / np.sqrt(deg), rtol=1e-7)


def test_gcn_prepare_grouped_matches_generic():
    """The sort-free build for frontier-ordered edge lists (GRAPES_PREP_SRC_GROUPED) must give the very
    same CSRs as the generic build — hub sources (>> GRAPES_LONG_ROW), self-loops, empty rows."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(17)
    n = 30000
    hub = np.stack([np.full(9000, 7, np.int64), rng.permutation(n)[:9000]])
    rnd = rng.integers(0, n, (2, 150000))
    loops = np.stack([np.arange(0, n, 3), np.arange(0, n, 3)])
    indptr, indices = O.build_csr(np.concatenate([hub, hub[::-1], rnd, rnd[::-1], loops], axis=1), n)
    prev = rng.permutation(n)[:600].astype(np.int64)
    prev[0] = 7
    tm = O.TensorMap(n)
    nb, batch_nodes, _, local = O.hop_index_pipeline(prev, indptr, indices, tm, n)
    assert (local[0] == local[1]).any()                       # the list does contain self-loops
    nloc = len(batch_nodes)
    ls, ld = _t(local[0], torch.int32), _t(local[1], torch.int32)
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    a = ops.PreparedGraph(ls, ld, nloc, status=st)
    b = ops.PreparedGraph(ls, ld, nloc, status=st, src_grouped=True)
    assert int(st.item()) == 0
    ne = int(a.rowptr_t[nloc].item())
    for name in ("rowptr_t", "rowptr_s", "dinv"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert torch.equal(a.csr_src[:ne], b.csr_src[:ne]) and torch.equal(a.csr_dst[:ne], b.csr_dst[:ne])
    assert a.n_long.tolist()[:3] == b.n_long.tolist()[:3] and b.n_long.tolist()[1] >= 9000 // 64 and b.n_long.tolist()[2] == ne
    # an edge list that is NOT grouped must be flagged, not silently mis-built
    perm = rng.permutation(local.shape[1])
    ops.PreparedGraph(_t(local[0][perm], torch.int32), _t(local[1][perm], torch.int32), nloc, status=st, src_grouped=True)
    assert int(st.item()) & 4


@pytest.mark.parametrize("n_prev,with_map", [(40, False), (40, True), (600, True), (12, True)])
def test_gcn_prepare_small_graph_and_in_kernel_relabel(n_prev, with_map):
    """The one-workgroup build for graphs of <= 2048 nodes and the in-kernel TensorMap relabel (main.py:195,254)
    against the generic multi-launch build on explicitly relabelled edges: hub targets (> 64 and > 8 entries per row),
    self-loops, empty rows."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(23 + n_prev)
    n = 30000
    hub = np.stack([rng.permutation(n)[:700], np.full(700, 7, np.int64)])          # node 7 has in-degree ~700
    rnd = rng.integers(0, n, (2, 20000 if n_prev < 100 else 150000))
    loops = np.stack([np.arange(0, n, 3), np.arange(0, n, 3)])
    indptr, indices = O.build_csr(np.concatenate([hub, hub[::-1], rnd, rnd[::-1], loops], axis=1), n)
    prev = rng.permutation(n)[:n_prev].astype(np.int64)
    prev[0] = 7
    tm = O.TensorMap(n)
    nb, batch_nodes, _, local = O.hop_index_pipeline(prev, indptr, indices, tm, n)
    nloc = len(batch_nodes)
    assert (nloc <= 2048) == (n_prev < 100)                   # small path for the small cases, general path otherwise
    glob = np.stack([batch_nodes[local[0]], batch_nodes[local[1]]])                  # the same edges with global ids
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    ref = ops.PreparedGraph(_t(local[0], torch.int32), _t(local[1], torch.int32), nloc, status=st)       # generic build
    if with_map:
        node_map = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        node_map[_t(batch_nodes).long()] = torch.arange(nloc, dtype=torch.int32, device="cuda")
        got = ops.PreparedGraph(_t(glob[0], torch.int32), _t(glob[1], torch.int32), nloc, status=st, src_grouped=True,
                                node_map=node_map)
    else:
        got = ops.PreparedGraph(_t(local[0], torch.int32), _t(local[1], torch.int32), nloc, status=st, src_grouped=True)
    assert int(st.item()) == 0
    ne = int(ref.rowptr_t[nloc].item())
    for name in ("rowptr_t", "rowptr_s", "dinv"):
        assert torch.equal(getattr(ref, name), getattr(got, name)), name
    assert torch.equal(ref.csr_src[:ne], got.csr_src[:ne]) and torch.equal(ref.csr_dst[:ne], got.csr_dst[:ne])
    assert ref.n_long.tolist()[:3] == got.n_long.tolist()[:3] and got.n_long.tolist()[2] == ne
    for half, cnt in ((0, got.n_long.tolist()[0]), (1, got.n_long.tolist()[1])):     # same work items, any slot order
        a = ref.long_items.view(2, -1, 2)[half, :cnt].cpu().numpy(); b = got.long_items.view(2, -1, 2)[half, :cnt].cpu().numpy()
        assert sorted(map(tuple, a)) == sorted(map(tuple, b))
    # device-side counts with capacity-sized buffers (the captured step's calling convention)
    cap_e, cap_n = local.shape[1] + 300, nloc + 50
    pad = lambda v, c: _t(np.concatenate([v, np.full(c - len(v), 5)]), torch.int32)
    d_e = torch.tensor([local.shape[1]], dtype=torch.int32, device="cuda"); d_n = torch.tensor([nloc], dtype=torch.int32, device="cuda")
    if with_map:
        cap = ops.PreparedGraph(pad(glob[0], cap_e), pad(glob[1], cap_e), cap_n, d_n=d_n, d_e=d_e, status=st, src_grouped=True,
                                node_map=node_map)
    else:
        cap = ops.PreparedGraph(pad(local[0], cap_e), pad(local[1], cap_e), cap_n, d_n=d_n, d_e=d_e, status=st, src_grouped=True)
    assert int(st.item()) == 0
    assert torch.equal(cap.rowptr_t[:nloc + 1], ref.rowptr_t) and torch.equal(cap.rowptr_s[:nloc + 1], ref.rowptr_s)
    assert torch.equal(cap.csr_src[:ne], ref.csr_src[:ne]) and torch.equal(cap.csr_dst[:ne], ref.csr_dst[:ne])
    assert torch.equal(cap.dinv[:nloc], ref.dinv)
    # a list that is not grouped is flagged by the small path too
    if nloc <= 2048:
        perm = rng.permutation(local.shape[1])
        ops.PreparedGraph(_t(local[0][perm], torch.int32), _t(local[1][perm], torch.int32), nloc, status=st, src_grouped=True)
        assert int(st.item()) & 4


@pytest.mark.parametrize("n_prev", [40, 600])
def test_gather_spmm_from_row_heads_is_bit_identical(n_prev):
    """The fused gather-SpMM driven by gcn_prepare's per-row head records (two dependent round trips) against the
    CSR-walking kernel and against the oracle's gcn_conv on the gathered features: rows with 0, 1-4, 5-8 and hundreds
    of entries; general and one-workgroup builds."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(31 + n_prev)
    n, F, num_ind = 30000, 100, 4
    hub = np.stack([rng.permutation(n)[:700], np.full(700, 7, np.int64)])
    rnd = rng.integers(0, n, (2, 20000 if n_prev < 100 else 150000))
    indptr, indices = O.build_csr(np.concatenate([hub, hub[::-1], rnd, rnd[::-1]], axis=1), n)
    prev = rng.permutation(n)[:n_prev].astype(np.int64); prev[0] = 7
    tm = O.TensorMap(n)
    nb, batch_nodes, _, local = O.hop_index_pipeline(prev, indptr, indices, tm, n)
    nloc = len(batch_nodes)
    # a frontier list carries each edge once (source = queried node); add the reverse direction so that by-target rows
    # of every length occur (the hub row gets hundreds of entries)
    rev = (rng.random(local.shape[1]) < 0.3) | (local[0] == int(np.searchsorted(batch_nodes, 7)))   # the hub's edges + 30 %
    ls = np.concatenate([local[0], local[1][rev]]); ld = np.concatenate([local[1], local[0][rev]])
    order = np.lexsort((ld, ls)); ls, ld = ls[order], ld[order]
    X = _t(rng.standard_normal((n, F)).astype(np.float32))
    ids = _t(batch_nodes, torch.int32)
    epoch = 9
    code = _t(((epoch << 8) | rng.integers(0, 16, n)).astype(np.int32))
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    plain = ops.PreparedGraph(_t(ls, torch.int32), _t(ld, torch.int32), nloc, status=st, src_grouped=True)
    heads = ops.PreparedGraph(_t(ls, torch.int32), _t(ld, torch.int32), nloc, status=st, src_grouped=True, head_ids=ids)
    assert int(st.item()) == 0 and heads.row_head is not None
    lens = (plain.rowptr_t[1:] - plain.rowptr_t[:-1]).cpu().numpy()
    assert lens.max() > 64 and (lens == 0).any() and (n_prev < 100 or ((lens > 4) & (lens <= 8)).any())
    hd = heads.row_head.cpu().numpy()
    assert np.array_equal(hd[:, 0], lens) and np.array_equal(hd[:, 1], batch_nodes)
    a = ops.gcn_aggregate_gather(X, ids, plain, code, epoch, num_ind)
    b = ops.gcn_aggregate_gather(X, ids, heads, code, epoch, num_ind)
    assert torch.equal(a, b)
    xg = ops.gather_rows(X, ids, code, epoch, num_ind)
    ref = O.gcn_conv(xg.cpu(), torch.eye(F + num_ind), None, torch.from_numpy(np.stack([ls, ld])))   # Â [X | ind]: W = I, no bias
    assert float((b.cpu() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    # the same rows read through a table of row shards (peer.PeerFeatures: 1, 3 and 8 shards of uneven size, one of them empty,
    # each its own allocation) — short rows, 5..16-entry rows and the hub row all pick the shard per row: bit-identical
    from grapes_amd.peer import PeerFeatures
    for cuts in ([0, n], [0, 11, 17000, n], [0, 3000, 3000, 9000, 12000, 20000, 25000, 29999, n]):
        pf = PeerFeatures.from_shards([X[a:z].clone() for a, z in zip(cuts, cuts[1:])])
        assert torch.equal(ops.gcn_aggregate_gather(pf, ids, heads, code, epoch, num_ind), b), cuts
        pick = _t(rng.integers(0, n, 40), torch.int64)
        assert torch.equal(pf.rows_for_check(pick), X[pick])


def test_gcn_module_layerwise_routing_and_state_dict():
    _cuda()
    from grapes_amd.modules.gcn import GCN
    rng = np.random.default_rng(9)
    n, F, H, C = 700, 100, 256, 47
    torch.manual_seed(3)
    ref = O.GCNRef(F, [H, H, C])
    net = GCN(F, [H, H, C]).cuda()
    assert list(net.state_dict().keys()) == list(ref.state_dict().keys())            # gcn_layers.{i}.lin.weight / .bias
    net.load_state_dict(ref.state_dict())
    eis = [_rand_edges(rng, n, 900), _rand_edges(rng, n, 1500), _rand_edges(rng, n, 700)]
    x = torch.randn(n, F)
    out, mem = net(x.cuda(), [torch.from_numpy(e).cuda() for e in eis])
    rout, _ = ref(x, [torch.from_numpy(e) for e in eis])                                # gcn.py:30-36 routing
    assert isinstance(mem, float)
    assert _close(out.detach().cpu().numpy(), rout.detach().numpy())
    y = torch.from_numpy(rng.integers(0, C, n))
    torch.nn.functional.cross_entropy(out, y.cuda()).backward()
    torch.nn.functional.cross_entropy(rout, y).backward()
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert _close(p.grad.cpu().numpy(), q.grad.numpy(), 2e-5), k
    # single edge_index for all layers (eval.py:50 full-batch form)
    out1, _ = net(x.cuda(), torch.from_numpy(eis[1]).cuda())
    rout1, _ = ref(x, torch.from_numpy(eis[1]))
    assert _close(out1.detach().cpu().numpy(), rout1.detach().numpy())


# ------------------------------------------------------------------------------ A8 whole step
@pytest.mark.parametrize("tag", ["small", "mid"])
def test_step_index_pipeline_golden(golden_dir, tag):
    """main.py:157-256 with injected logits against the reference-generated trace (G5): every index
    array bit-exact."""
    _cuda()
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.step import GrapesTrainer
    g = _load(golden_dir, "g5_step_trace.npz")
    n, B, K, hops = [int(v) for v in g[f"{tag}_cfg"]]
    dg = DeviceGraph.from_csr(g[f"{tag}_indptr"], g[f"{tag}_indices"])
    X = torch.zeros(n, 4, device="cuda")

    def inject(hop, batch_nodes):      # evaluated on the CPU exactly as the fixture generator did
        v = batch_nodes.cpu().to(torch.float64)
        return (3.0 * torch.sin(0.37 * v + 1.3 * hop)).to(torch.float32).cuda()

    tr = GrapesTrainer(dg, X, None, None, None, None, sampling_hops=hops, num_samples=K)
    out = tr.step(torch.from_numpy(g[f"{tag}_targets"]), uniforms_fn=lambda hop, nn: _t(g[f"{tag}_h{hop}_uniforms"]),
                  inject_logits_fn=inject, trace=True)
    c = lambda t: t.cpu().numpy().astype(np.int64)
    for hop in range(hops):
        p, h = f"{tag}_h{hop}_", out["hops"][hop]
        assert np.array_equal(c(h["neighborhoods"]), g[p + "neigh"])
        assert np.array_equal(c(h["batch_nodes"]), g[p + "batch_nodes"])
        assert np.array_equal(c(h["neighbor_nodes"]), g[p + "neighbor_nodes"])
        assert np.array_equal(c(h["local_neighborhoods"]), g[p + "local"])
        assert np.array_equal(h["indicator_rows"].cpu().numpy(), g[p + "ind_rows"])
        assert np.array_equal(c(h["kept"]), g[p + "kept"])
        assert np.array_equal(c(h["k_hop_edges"]), g[p + "k_hop_edges"])
        assert np.allclose(h["log_prob"].cpu().numpy(), g[p + "logp"], rtol=1e-5, atol=1e-6)
    assert np.array_equal(c(out["all_nodes"]), g[f"{tag}_all_nodes"])
    for i in range(hops):
        assert np.array_equal(c(out["edge_indices"][i]), g[f"{tag}_edge_index_{i}"])
    assert np.array_equal(c(out["local_target_ids"]), g[f"{tag}_local_targets"])


@pytest.mark.parametrize("cfg", ["tiny", "arxiv_like"])
def test_full_training_step_vs_oracle(cfg):
    """Whole iteration (sampler GCN, draw, log-Z net, classifier, both losses, both backward passes)
    against the CPU oracle on identical weights, targets and uniforms."""
    _cuda()
    from grapes_amd import synth
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.step import GrapesTrainer
    if cfg == "tiny":
        n, deg, F, C, B, K, hops, H = 400, 6.0, 33, 5, 24, 12, 2, 32
    else:
        n, deg, F, C, B, K, hops, H = 20000, 13.7, 128, 40, 256, 256, 3, 256
    indptr, indices = synth.synth_csr_numpy(n, deg, 500, seed=1)
    rng = np.random.default_rng(2)
    X = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, C, n))
    targets = rng.permutation(n)[:B].astype(np.int64)
    torch.manual_seed(0)
    dims_c = [H] * (hops - 1) + [C]
    ref_c, ref_gf, ref_z = O.GCNRef(F, dims_c), O.GCNRef(F + hops + 1, [H, 1]), O.GCNRef(F, [H, 1])
    c, gf, z = GCN(F, dims_c).cuda(), GCN(F + hops + 1, [H, 1]).cuda(), GCN(F, [H, 1]).cuda()
    c.load_state_dict(ref_c.state_dict()); gf.load_state_dict(ref_gf.state_dict()); z.load_state_dict(ref_z.state_dict())
    uni = {h: rng.random(n, dtype=np.float32) for h in range(hops)}
    coef = 10.0
    ot = O.train_step(indptr, indices, X, y, targets, ref_c, ref_gf, ref_z, sampling_hops=hops, num_samples=K,
                      uniforms_fn=lambda h, nn: uni[h][:nn], loss_coef=coef)
    dg = DeviceGraph.from_csr(indptr, indices)
    tr = GrapesTrainer(dg, X.cuda(), y.cuda(), c, gf, z, sampling_hops=hops, num_samples=K, loss_coef=coef)
    out = tr.step(torch.from_numpy(targets), uniforms_fn=lambda h, nn: _t(uni[h][:nn]), trace=True)
    ci = lambda t: t.cpu().numpy().astype(np.int64)
    for hop in range(hops):
        h, oh = out["hops"][hop], ot["hops"][hop]
        assert np.array_equal(ci(h["batch_nodes"]), oh["batch_nodes"]), hop
        assert _close(h["cand_logits"].cpu().numpy(), oh["cand_logits"].numpy()), hop        # layer activations 1e-5
        assert np.array_equal(ci(h["kept"]), oh["kept"]), hop                                # sampled set bit-exact
        assert np.array_equal(ci(h["k_hop_edges"]), oh["k_hop_edges"]), hop
    assert np.array_equal(ci(out["all_nodes"]), ot["all_nodes"])
    assert _close(out["logits"].cpu().numpy(), ot["logits"].numpy())
    assert abs(float(out["loss_c"]) - ot["loss_c"]) <= 1e-5 * max(1, abs(ot["loss_c"]))
    assert abs(float(out["log_z"]) - ot["log_z"]) <= 1e-5 * max(1, abs(ot["log_z"]))
    assert abs(float(out["tot_log_prob"]) - ot["tot_log_prob"]) <= 2e-5 * max(1, abs(ot["tot_log_prob"]))
    assert abs(float(out["loss_gfn"]) - ot["loss_gfn"]) <= 1e-4 * max(1, abs(ot["loss_gfn"]))
    assert GrapesTrainer.edges_aggregated(out) == ot["edges_aggregated"]
    for name, net, ref in (("c", c, ref_c), ("gf", gf, ref_gf), ("z", z, ref_z)):
        for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
            scale = max(1.0, float(q.grad.abs().max()))
            assert float((p.grad.cpu() - q.grad).abs().max()) <= 1e-4 * scale, (name, k)


@pytest.mark.parametrize("capture,reinforce,forms", [(False, False, "default"), (True, False, "default"), (True, True, "default"),
                                                     (True, False, "activations"), (True, False, "bits, separate launches"),
                                                     (True, False, "1 hop"), (True, False, "4 hops")])
def test_captured_step_matches_eager_step(capture, reinforce, forms, monkeypatch):
    """step_graph.GraphedTrainer (sync-free, explicit backward, one hipGraph per iteration) against
    step.GrapesTrainer (exact-size tensors + autograd) over several consecutive training iterations with
    Adam: identical sampled sets every step, losses / weights within fp32 tolerance."""
    _cuda()
    from grapes_amd import synth
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.step import GrapesTrainer
    from grapes_amd.step_graph import GraphedTrainer
    # the hidden layers of the sampler / log-Z nets: gate bits + paired backward launch (default), the activation forms, or
    # gate bits with one backward launch per net
    if forms == "activations":
        monkeypatch.setenv("GRAPES_DIAG", "1")          # (Python-side switches are read only in a diagnostic session)
        monkeypatch.setenv("GRAPES_GATE_BITS", "0")
    elif forms == "bits, separate launches":
        monkeypatch.setenv("GRAPES_DIAG", "1")
        monkeypatch.setenv("GRAPES_DW_PAIR", "0")
    n, deg, F, C, B, K, hops, H = 30000, 12.0, 100, 9, 128, 96, 3, 256
    if forms.endswith("hop") or forms.endswith("hops"):      # one hop: the log-Z net rides with a single row set; four: it cannot
        hops = int(forms.split()[0])
    indptr, indices = synth.synth_csr_numpy(n, deg, 2000, seed=11)
    rng = np.random.default_rng(12)
    X = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.integers(0, C, n)).cuda()
    batches = [torch.from_numpy(rng.permutation(n)[:B].astype(np.int64)).cuda() for _ in range(5)]

    def build():
        torch.manual_seed(0)
        c, gf, z = GCN(F, [H] * (hops - 1) + [C]).cuda(), GCN(F + hops + 1, [H, 1]).cuda(), GCN(F, [H, 1]).cuda()
        oc = torch.optim.Adam(c.parameters(), lr=1e-3, capturable=True)
        og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=1e-4, capturable=True)
        return c, gf, z, oc, og

    c, gf, z, oc, og = build()
    eager = GrapesTrainer(DeviceGraph.from_csr(indptr, indices), X, y, c, gf, z, sampling_hops=hops, num_samples=K,
                          loss_coef=50.0, optimizer_c=oc, optimizer_gf=og, philox_seed=77, reinforce_baseline=reinforce)
    c2, gf2, z2, oc2, og2 = build()
    graphed = GraphedTrainer(DeviceGraph.from_csr(indptr, indices), X, y, c2, gf2, z2, batch_size=B, sampling_hops=hops,
                             num_samples=K, loss_coef=50.0, optimizer_c=oc2, optimizer_gf=og2, e_cap=1 << 15,
                             philox_seed=77, capture=capture, reinforce_baseline=reinforce)   # main.py:279 when True
    for it, tg in enumerate(batches):
        a = eager.step(tg, trace=True)
        b = graphed.step(tg)
        torch.cuda.synchronize()
        graphed.check()
        for hop in range(hops):
            ka = a["hops"][hop]["kept"]
            kc = int(b["kept_counts"][hop].item())
            assert kc == ka.numel(), (it, hop)
            assert torch.equal(b["kept"][hop][:kc], ka.to(torch.int32)), (it, hop)          # sampled sets bit-exact
        na = int(b["n_all"].item())
        assert torch.equal(b["all_nodes"][:na], a["all_nodes"])
        # it == 0: identical weights => activations within 1e-5 (in fact bit-equal: same kernels).  Later iterations
        # start from weights updated from two loss-gradient implementations (torch autograd's CrossEntropy backward vs
        # the fused loss kernel) that differ by ~1e-9 absolute: Adam's update lr*g/(|g|+eps) has slope
        # lr*eps/(|g|+eps)^2 ~ 1e3..1e4 where |g| ~ eps, which moves a handful of weights by ~1e-6 and logits by ~1e-5.
        tol = dict(rtol=1e-5, atol=1e-6) if it == 0 else dict(rtol=2e-4, atol=3e-5)
        assert torch.allclose(b["logits"][:na], a["logits"], **tol), it
        assert abs(float(b["loss_c"]) - float(a["loss_c"])) <= 1e-5 * max(1.0, abs(float(a["loss_c"])))
        assert abs(float(b["log_z"]) - float(a["log_z"])) <= 1e-5 * max(1.0, abs(float(a["log_z"])))
        assert abs(float(b["tot_log_prob"]) - float(a["tot_log_prob"])) <= 2e-5 * max(1.0, abs(float(a["tot_log_prob"])))
        assert abs(float(b["loss_gfn"]) - float(a["loss_gfn"])) <= 1e-4 * max(1.0, abs(float(a["loss_gfn"])))
        assert GraphedTrainer.edges_aggregated(b) == GrapesTrainer.edges_aggregated(a)
    for m1, m2 in ((c, c2), (gf, gf2), (z, z2)):
        for (k, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
            assert torch.allclose(p, q, rtol=1e-4, atol=1e-5), k
    if capture:
        assert graphed.graph_obj is not None
    # capacity overflow is reported, never silent
    small = GraphedTrainer(DeviceGraph.from_csr(indptr, indices), X, y, c2, gf2, z2, batch_size=B, sampling_hops=hops,
                           num_samples=K, loss_coef=50.0, e_cap=256, philox_seed=1, capture=False)
    small.step(batches[0])
    from grapes_amd import _lib
    with pytest.raises(_lib.GrapesHipError):
        small.check()


@pytest.mark.parametrize("n,cap,K,H,strided", [(37500, 37500, 104, 256, False), (5000, 9000, 104, 256, False),
                                                 (4099, 4099, 100, 256, True), (2500, 2500, 64, 96, False),
                                                 # f_in > 112 (arxiv / papers100M: 128 features + indicators): the 160-column form
                                                 (37000, 37000, 132, 256, False), (5000, 9000, 128, 256, False),
                                                 (3000, 3000, 156, 128, False)])
def test_gate_bits_layer_pair_matches_activation_form(n, cap, K, H, strided):
    """layer -> ReLU -> 1-wide head with 32 bytes of gate bits per row instead of the activations (include/grapes_hip.h):
    same head output bit for bit, bits = (activation > 0) in the documented layout, and dW1 / db1 / dW2 against the
    activation-based kernels and against fp64 torch autograd of the same expression (modules/gcn.py:31-36, [H, 1])."""
    _cuda()
    from grapes_amd import ops
    torch.manual_seed(n + K)
    wide = torch.randn(cap, K + 4 if strided else K, device="cuda")
    x = wide[:, :K]
    w = (torch.randn(H, K, device="cuda") * 0.2).contiguous()
    b = torch.randn(H, device="cuda") * 0.1
    w2 = torch.randn(1, H, device="cuda") * 0.3
    d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    if not ops.split_gemm_available(cap, K, H):
        pytest.skip("bf16x3 kernels not available for this shape")
    if strided:
        act, head = ops.linear_bias_act_head_fwd_strided(x, w, b, True, w2, d_n=d_n)
    else:
        act, head = ops.linear_bias_act_head_fwd(x.contiguous(), w, b, True, w2, d_n=d_n)
    bits, head_b = ops.linear_relu_head_fwd_bits(x, w, b, w2, d_n=d_n)
    assert torch.equal(head[:n], head_b[:n])
    # decode: bit 16 h + 4 q + u of word [r][c // 32]  <->  column 32 (c // 32) + 8 q + 4 h + u
    words = bits.words[:n].cpu().numpy().astype(np.uint32)
    dec = np.zeros((n, H), dtype=bool)
    for wv in range(H // 32):
        for q in range(4):
            for hh in range(2):
                for u in range(4):
                    dec[:, 32 * wv + 8 * q + 4 * hh + u] = (words[:, wv] >> (16 * hh + 4 * q + u)) & 1
    assert np.array_equal(dec, (act[:n] > 0).cpu().numpy())
    rs = torch.randn(cap, device="cuda")
    outs = []
    wide_k = K > 112          # only the gate-word backward has the 160-column form: checked against fp64 autograd alone
    for form in ("act", "bits"):
        dw = torch.full((H, K), 7.0, device="cuda"); db = torch.full((H,), 7.0, device="cuda"); dwh = torch.full((H,), 7.0, device="cuda")
        if form == "act" and wide_k:
            outs.append(None)
            continue
        if form == "bits":
            ops.linear_bwd_weight_bits_multi([bits], [x], [rs], [d_n], w2.view(-1), w, b, dw, dbias=db, dw_head=dwh)
        elif strided:
            ops.linear_bwd_weight_gated_strided(x, act, rs, w2.view(-1), dw, dbias=db, dw_head=dwh, d_n=d_n)
        else:
            ops.linear_bwd_weight_gated(None, x.contiguous(), gate=act, d_n=d_n, dw=dw, dbias=db, row_scale=rs, col_vec=w2.view(-1),
                                        dw_head=dwh)
        outs.append((dw, db, dwh))
    # same MFMA sequence on the same mask; the head's weight scales a slab in one form and the slab sum in the other
    if not wide_k:
        assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-5 * float(outs[0][0].abs().max()))
        assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-5 * float(outs[0][1].abs().max()))
    xd, wd, bd, w2d = x[:n].double().requires_grad_(False), w.double().requires_grad_(True), b.double().requires_grad_(True), \
        w2.double().requires_grad_(True)
    hd = torch.relu(xd @ wd.t() + bd) @ w2d.t()
    hd.backward(rs[:n].double().view(-1, 1))
    for got, ref in ((outs[1][0], wd.grad), (outs[1][1], bd.grad), (outs[1][2], w2d.grad.view(-1))) + \
            (() if wide_k else ((outs[0][2], w2d.grad.view(-1)),)):
        scale = max(1.0, float(ref.abs().max()))
        assert float((got.double() - ref).abs().max()) <= 2e-5 * scale
    # two row sets sharing the weights, accumulated on top of existing gradients
    n2 = n // 3
    d_n2 = torch.tensor([n2], dtype=torch.int32, device="cuda")
    x2 = torch.randn(n2 + 5, K, device="cuda")
    bits2, _ = ops.linear_relu_head_fwd_bits(x2, w, b, w2, d_n=d_n2) if ops.split_gemm_available(n2 + 5, K, H) else (None, None)
    if bits2 is not None:
        rs2 = torch.randn(n2 + 5, device="cuda")
        dw = outs[1][0].clone(); db = outs[1][1].clone(); dwh = outs[1][2].clone()
        ops.linear_bwd_weight_bits_multi([bits2, bits], [x2, x], [rs2, rs], [d_n2, d_n], w2.view(-1), w, b, dw, dbias=db,
                                         dw_head=dwh, accumulate=True)
        for t in (wd, bd, w2d):
            t.grad = None
        h1 = torch.relu(xd @ wd.t() + bd) @ w2d.t()
        h2 = torch.relu(x2[:n2].double() @ wd.t() + bd) @ w2d.t()
        (2.0 * (h1 * rs[:n].double().view(-1, 1)).sum() + (h2 * rs2[:n2].double().view(-1, 1)).sum()).backward()
        for got, ref in ((dw, wd.grad), (db, bd.grad), (dwh, w2d.grad.view(-1))):
            assert float((got.double() - ref).abs().max()) <= 4e-5 * max(1.0, float(ref.abs().max()))
    # a second layer over the leading columns of the same rows (the log-Z net beside the sampler net) as the second problem
    # of ONE launch: equal to two separate launches up to the summation order of the row shares
    Kb = K - 4
    if ops.split_gemm_available(cap, Kb, H):
        wb = (torch.randn(H, Kb, device="cuda") * 0.2).contiguous(); bb = torch.randn(H, device="cuda") * 0.1
        w2b = torch.randn(1, H, device="cuda") * 0.3; rsb = torch.randn(cap, device="cuda")
        xb = x[:, :Kb]
        bits_b, _ = ops.linear_relu_head_fwd_bits(xb, wb, bb, w2b, d_n=d_n)
        sep = [torch.zeros(H, K, device="cuda"), torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda"),
               torch.zeros(H, Kb, device="cuda"), torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")]
        ops.linear_bwd_weight_bits_multi([bits], [x], [rs], [d_n], w2.view(-1), w, b, sep[0], dbias=sep[1], dw_head=sep[2])
        ops.linear_bwd_weight_bits_multi([bits_b], [xb], [rsb], [d_n], w2b.view(-1), wb, bb, sep[3], dbias=sep[4], dw_head=sep[5])
        pair = [torch.full_like(t, 3.0) for t in sep]
        ops.linear_bwd_weight_bits_pair([bits], [x], [rs], [d_n], w2.view(-1), w, b, pair[0], pair[1], pair[2],
                                        bits_b, xb, rsb, d_n, w2b.view(-1), wb, bb, pair[3], pair[4], pair[5])
        for got, ref in zip(pair, sep):
            assert torch.allclose(got, ref, rtol=1e-5, atol=2e-5 * max(1.0, float(ref.abs().max())))


def test_shared_launch_forms_equal_the_separate_launches():
    """The launches that carry several problems at once against one launch per problem: the few-row weight gradients of three
    layers with ONE slab reduction (ops.DeferredSlabs) and the two 1-wide heads aggregated over one graph in one launch
    (ops.gcn_aggregate_narrow_pair) — bit-equal, they run the same arithmetic in the same order."""
    _cuda()
    from grapes_amd import ops
    torch.manual_seed(5)
    n, cap = 1022, 1025
    d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    shapes = [(47, 256, False), (256, 256, False), (256, 100, True)]      # (f_out, f_in, gated + bias sum): the classifier's layers
    ops_in = []
    for fo, fi, gated in shapes:
        dh = torch.randn(cap, fo, device="cuda"); x = torch.randn(cap, fi, device="cuda")
        gate = torch.randn(cap, fo, device="cuda") if gated else None
        ops_in.append((dh, x, gate))
    sep, defd = [], []
    deferred = ops.DeferredSlabs()
    for (dh, x, gate), (fo, fi, gated) in zip(ops_in, shapes):
        if gated:
            a = ops.linear_bwd_weight_gated(dh, x, gate=gate, d_n=d_n, dw=torch.full((fo, fi), 2.0, device="cuda"),
                                            dbias=torch.full((fo,), 2.0, device="cuda"))
            b = ops.linear_bwd_weight_gated(dh, x, gate=gate, d_n=d_n, dw=torch.full((fo, fi), 3.0, device="cuda"),
                                            dbias=torch.full((fo,), 3.0, device="cuda"), defer=deferred)
            sep += list(a); defd += list(b)
        else:
            sep.append(ops.linear_bwd_weight(dh, x, d_n=d_n, out=torch.full((fo, fi), 2.0, device="cuda")))
            defd.append(ops.linear_bwd_weight(dh, x, d_n=d_n, out=torch.full((fo, fi), 3.0, device="cuda"), defer=deferred))
    assert len(deferred.sets) == 4                    # three weight gradients + one bias sum wait for the flush
    deferred.flush()
    for a, b in zip(sep, defd):
        assert torch.equal(a, b)
    ref = ops_in[2][0][:n].double() * (ops_in[2][2][:n] > 0)
    assert torch.allclose(defd[2].double(), ref.t() @ ops_in[2][1][:n].double(), rtol=1e-5, atol=1e-3)
    assert torch.allclose(defd[3].double(), ref.sum(0), rtol=1e-5, atol=1e-3)
    # a layer's weight and input gradient as one launch (two independent few-row GEMMs side by side)
    for fo, fi in ((47, 256), (256, 256)):
        dh = torch.randn(cap, fo, device="cuda"); x = torch.randn(cap, fi, device="cuda"); w = torch.randn(fo, fi, device="cuda")
        dfr = ops.DeferredSlabs()
        dw_p = torch.full((fo, fi), 3.0, device="cuda")
        dx_p = ops.linear_bwd_weight_and_input(dh, x, w, d_n=d_n, out=dw_p, defer=dfr)
        assert len(dfr.sets) == 1                     # the pair launch took it (not the two-call fallback + its own add)
        dfr.flush()
        assert torch.equal(dw_p, ops.linear_bwd_weight(dh, x, d_n=d_n))
        assert torch.equal(dx_p[:n], ops.linear_bwd_input(dh, w, d_n=d_n)[:n])
        assert torch.allclose(dx_p[:n].double(), dh[:n].double() @ w.double(), rtol=1e-5, atol=1e-3)
    # two heads over one graph
    rng = np.random.default_rng(6)
    nn_, e = 5000, 30000
    src = _t(np.sort(rng.integers(0, nn_, e)), torch.int32); dst = _t(rng.integers(0, nn_, e), torch.int32)
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    prep = ops.PreparedGraph(src, dst, nn_, status=st, src_grouped=True)
    ha, hb = torch.randn(nn_, 1, device="cuda"), torch.randn(nn_, 1, device="cuda")
    ba, bb = torch.randn(1, device="cuda"), torch.randn(1, device="cuda")
    oa, ob = ops.gcn_aggregate_narrow_pair(ha, hb, prep, ba, bb)
    assert torch.equal(oa, ops.gcn_aggregate_fwd(ha, prep, ba, False)) and torch.equal(ob, ops.gcn_aggregate_fwd(hb, prep, bb, False))
    # the aggregation of a transform-first layer that also returns the 1-wide head's X W step: the same activations bit for
    # bit, the head within fp32 rounding of act @ w2 (fp64); hub rows (> 16 entries) and a live count below the capacity included
    d_n = torch.tensor([nn_ - 37], dtype=torch.int32, device="cuda")
    keep = (src < nn_ - 37) & (dst < nn_ - 37)
    prep_f = ops.PreparedGraph(src, dst, nn_, status=st, src_grouped=True, items_fwd=False)
    prep_n = ops.PreparedGraph(src[keep], dst[keep], nn_, d_n=d_n, status=st, src_grouped=True, items_fwd=False)
    for f in (256, 64, 20):
        h = torch.randn(nn_, f, device="cuda"); b1 = torch.randn(f, device="cuda"); w2 = torch.randn(f, device="cuda")
        for pg, live in ((prep_f, nn_), (prep_n, nn_ - 37)):
            r = ops.gcn_aggregate_fwd_head(h, pg, b1, True, w2)
            assert r is not None
            act = ops.gcn_aggregate_fwd(h, pg, b1, True)
            assert torch.equal(r[0][:live], act[:live])
            ref = act[:live].double() @ w2.double()
            mag = act[:live].double().abs() @ w2.double().abs()
            assert float(((r[1][:live, 0].double() - ref).abs() / (mag + 1e-30)).max()) <= 2e-6
    # ... and the ReLU gate bits of its output; the backward aggregation that reads them (32 bytes per row instead of the row)
    # returns the aggregation over the activation rows bit for bit, with and without long rows as work items, hub rows included
    extra = [(11, 3000), (12, 40), (13, 64), (14, 17), (15, 65), (16, 16)]     # (source, entries): work items, eight chains, one chain
    src2 = torch.cat([torch.full((k,), sid, dtype=torch.int32, device="cuda") for sid, k in extra] + [src[keep]])
    dst2 = torch.cat([_t(rng.integers(0, nn_ - 37, k), torch.int32) for _, k in extra] + [dst[keep]])
    order = torch.argsort(src2.long(), stable=True)
    preps = [prep_n, ops.PreparedGraph(src2[order], dst2[order], nn_, d_n=d_n, status=st, src_grouped=True, items_fwd=False)]
    # a graph small enough to run without work items: its 200-entry row is walked inside the streaming kernel (64 entries a block)
    ns_ = 1500
    ssrc = torch.cat([torch.full((200,), 7, dtype=torch.int32, device="cuda"), _t(np.sort(rng.integers(8, ns_, 4000)), torch.int32)])
    sdst = _t(rng.integers(0, ns_, 4200), torch.int32)
    prep_s = ops.PreparedGraph(ssrc, sdst, ns_, status=st, src_grouped=True, items_fwd=False)
    hs_ = torch.randn(ns_, 256, device="cuda"); bs_ = torch.randn(256, device="cuda"); ws_ = torch.randn(256, device="cuda")
    outs, _, bits_s = ops.gcn_aggregate_fwd_head(hs_, prep_s, bs_, True, ws_, want_bits=True)
    dhs = torch.randn(ns_, device="cuda")
    assert torch.equal(ops.gcn_aggregate_bwd_rank1(outs, dhs, ws_, prep_s, gate_bits=bits_s), ops.gcn_aggregate_bwd_rank1(outs, dhs, ws_, prep_s))
    for nt in (1, 3, 6):                                  # fewer rows than one wavefront round holds
        ts_ = torch.zeros(nt + 1, dtype=torch.int32, device="cuda")[:nt]; td_ = torch.arange(nt, dtype=torch.int32, device="cuda")      # (source 0 -> every row, itself included)
        prep_t = ops.PreparedGraph(ts_.contiguous(), td_.contiguous(), nt, status=st, src_grouped=True, items_fwd=False)
        ht = torch.randn(nt, 64, device="cuda"); bt = torch.randn(64, device="cuda"); wt = torch.randn(64, device="cuda")
        ot, _, bits_t = ops.gcn_aggregate_fwd_head(ht, prep_t, bt, True, wt, want_bits=True)
        assert torch.equal(ot, ops.gcn_aggregate_fwd(ht, prep_t, bt, True))
        dt = torch.randn(nt, device="cuda")
        assert torch.equal(ops.gcn_aggregate_bwd_rank1(ot, dt, wt, prep_t, gate_bits=bits_t), ops.gcn_aggregate_bwd_rank1(ot, dt, wt, prep_t))
    for f in (256, 64, 20):
        h = torch.randn(nn_, f, device="cuda"); b1 = torch.randn(f, device="cuda"); w2 = torch.randn(f, device="cuda")
        dh2 = torch.randn(nn_, device="cuda")
        for pg in preps:
            live = nn_ - 37
            out, hw, bits = ops.gcn_aggregate_fwd_head(h, pg, b1, True, w2, want_bits=True)
            assert bits is not None and bits.shape == (nn_, 8)
            got = ((bits[:live].view(live, 8, 1) >> torch.arange(32, device="cuda", dtype=torch.int32).view(1, 1, 32)) & 1).reshape(live, 256)
            assert torch.equal(got[:, :f].bool(), out[:live] > 0) and int(got[:, f:].sum()) == 0
            dwa, dba = torch.zeros(f, device="cuda"), torch.zeros(f, device="cuda")
            dwb, dbb = torch.zeros(f, device="cuda"), torch.zeros(f, device="cuda")
            ref = ops.gcn_aggregate_bwd_rank1(out, dh2, w2, pg, dw_head=dwa, dbias=dba)
            new = ops.gcn_aggregate_bwd_rank1(out, dh2, w2, pg, dw_head=dwb, dbias=dbb, gate_bits=bits)
            assert torch.equal(new[:live], ref[:live]) and torch.equal(dwa, dwb) and torch.equal(dba, dbb)
    assert ops.gcn_aggregate_fwd_head(torch.randn(nn_, 8, device="cuda"), prep_f, None, True, torch.randn(8, device="cuda")) is None
    assert ops.gcn_aggregate_fwd_head(h, prep, b1, True, w2) is None          # (long rows as work items: the two-launch path)
    assert int(st.item()) == 0


def test_rank1_backward_of_three_independent_problems_in_five_launches():
    """grapes_gcn_aggregate_bwd_rank1_bits_multi: the rank-1 backward aggregations of two hops of one net (shared dW2 / db1: the second
    accumulates) and of a second net (its own outputs) in the SAME five launches — graphs of different sizes, with long rows as work
    items, without, and small enough to run without items — every dh, dW2 and db1 equal BIT FOR BIT to the calls one after the other;
    a first member that accumulates adds to what is there."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(16)
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    f = 256

    def graph(n, e, extra, live):
        src = np.sort(rng.integers(0, live, e)); dst = rng.integers(0, live, e)
        if extra:
            src = np.concatenate([np.concatenate([np.full(k, sid) for sid, k in extra]), src])
            dst = np.concatenate([np.concatenate([rng.integers(0, live, k) for _, k in extra]), dst])
            o = np.argsort(src, kind="stable"); src, dst = src[o], dst[o]
        d_n = torch.tensor([live], dtype=torch.int32, device="cuda")
        pg = ops.PreparedGraph(_t(src, torch.int32), _t(dst, torch.int32), n, d_n=d_n, status=st, src_grouped=True, items_fwd=False)
        h = torch.randn(n, f, device="cuda"); b1 = torch.randn(f, device="cuda"); w2 = torch.randn(f, device="cuda")
        out, _, bits = ops.gcn_aggregate_fwd_head(h, pg, b1, True, w2, want_bits=True)
        return dict(prep=pg, act=out, gate_bits=bits, w2=w2, dh2=torch.randn(n, device="cuda"), live=live)

    A = graph(23000, 60000, [(11, 3000), (12, 40), (13, 65)], 22950)
    B = graph(77000, 200000, [(5, 9000), (6, 17)], 76000)
    Cg = graph(1500, 4000, [(7, 200)], 1500)          # no work items
    B["w2"] = A["w2"]                                  # two hops of ONE net
    for trio in ((A, B, Cg), (B, A), (Cg, A, B)):
        seq_w, seq_b = [torch.full((f,), 0.5, device="cuda") for _ in range(2)], [torch.full((f,), -0.25, device="cuda") for _ in range(2)]
        mul_w, mul_b = [t.clone() for t in seq_w], [t.clone() for t in seq_b]
        # outputs: the problems with A's head weights share slot 0 (first writes or — last round — accumulates), the other takes slot 1
        def slots(p): return 0 if p["w2"] is A["w2"] else 1
        first_acc = trio[0] is Cg                      # (third round: the first member of its group accumulates into what is there)
        seen = set()
        accs = []
        for p in trio:
            k = slots(p)
            accs.append((k in seen) or (first_acc and p is Cg))
            seen.add(k)
        ref = [ops.gcn_aggregate_bwd_rank1(p["act"], p["dh2"], p["w2"], p["prep"], dw_head=seq_w[slots(p)], dbias=seq_b[slots(p)],
                                           accumulate=a, gate_bits=p["gate_bits"]) for p, a in zip(trio, accs)]
        got = ops.gcn_aggregate_bwd_rank1_multi([dict(act=p["act"], dh2=p["dh2"], w2=p["w2"], prep=p["prep"], gate_bits=p["gate_bits"],
                                                      dw_head=mul_w[slots(p)], dbias=mul_b[slots(p)], accumulate=a)
                                                 for p, a in zip(trio, accs)])
        torch.cuda.synchronize()
        for p, r, g in zip(trio, ref, got):
            assert torch.equal(g[:p["live"]], r[:p["live"]])
            assert float(r[:p["live"]].abs().sum()) > 0
        for k in range(2):
            assert torch.equal(seq_w[k], mul_w[k]) and torch.equal(seq_b[k], mul_b[k])
    # without column sums
    got = ops.gcn_aggregate_bwd_rank1_multi([dict(act=p["act"], dh2=p["dh2"], w2=p["w2"], prep=p["prep"], gate_bits=p["gate_bits"]) for p in (A, B)])
    for p, g in zip((A, B), got):
        assert torch.equal(g[:p["live"]], ops.gcn_aggregate_bwd_rank1(p["act"], p["dh2"], p["w2"], p["prep"], gate_bits=p["gate_bits"])[:p["live"]])
    assert int(st.item()) == 0


def test_random_sampling_step_vs_oracle_and_captured():
    """--random_sampling (reference configs/random/*, main.py:206-207,223,272): constant logits, uniform exact-k draw, no
    sampler / log-Z net, classifier update only.  The eager step against the CPU oracle on injected uniforms (sampled sets
    bit-exact), then the captured step against the eager one over several Adam iterations on the shared Philox stream."""
    _cuda()
    from grapes_amd import synth
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.step import GrapesTrainer
    from grapes_amd.step_graph import GraphedTrainer
    n, deg, F, C, B, K, hops, H = 6000, 9.0, 50, 7, 64, 48, 2, 64
    indptr, indices = synth.synth_csr_numpy(n, deg, 400, seed=21)
    rng = np.random.default_rng(22)
    X = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, C, n))
    targets = rng.permutation(n)[:B].astype(np.int64)
    torch.manual_seed(0)
    ref_c = O.GCNRef(F, [H, C])
    c = GCN(F, [H, C]).cuda(); c.load_state_dict(ref_c.state_dict())
    uni = {h: rng.random(n, dtype=np.float32) for h in range(hops)}
    ot = O.train_step(indptr, indices, X, y, targets, ref_c, None, None, sampling_hops=hops, num_samples=K,
                      uniforms_fn=lambda h, nn: uni[h][:nn], random_sampling=True)
    tr = GrapesTrainer(DeviceGraph.from_csr(indptr, indices), X.cuda(), y.cuda(), c, None, None, sampling_hops=hops,
                       num_samples=K, random_sampling=True)
    out = tr.step(torch.from_numpy(targets), uniforms_fn=lambda h, nn: _t(uni[h][:nn]), trace=True)
    ci = lambda t: t.cpu().numpy().astype(np.int64)
    for hop in range(hops):
        assert np.array_equal(ci(out["hops"][hop]["kept"]), ot["hops"][hop]["kept"]), hop
        assert np.array_equal(ci(out["hops"][hop]["k_hop_edges"]), ot["hops"][hop]["k_hop_edges"]), hop
    assert np.array_equal(ci(out["all_nodes"]), ot["all_nodes"])
    assert _close(out["logits"].cpu().numpy(), ot["logits"].numpy())
    assert abs(float(out["loss_c"]) - ot["loss_c"]) <= 1e-5 * max(1, abs(ot["loss_c"]))
    assert out.get("loss_gfn") is None and ot.get("loss_gfn") is None
    for (k, p_), (_, q) in zip(c.named_parameters(), ref_c.named_parameters()):
        assert float((p_.grad.cpu() - q.grad).abs().max()) <= 1e-4 * max(1.0, float(q.grad.abs().max())), k

    def build():
        torch.manual_seed(1)
        m = GCN(F, [H, C]).cuda()
        return m, torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True)

    Xd, yd = X.cuda(), y.cuda()
    batches = [torch.from_numpy(rng.permutation(n)[:B].astype(np.int64)).cuda() for _ in range(5)]
    m1, o1 = build()
    eager = GrapesTrainer(DeviceGraph.from_csr(indptr, indices), Xd, yd, m1, None, None, sampling_hops=hops, num_samples=K,
                          optimizer_c=o1, philox_seed=5, random_sampling=True)
    m2, o2 = build()
    graphed = GraphedTrainer(DeviceGraph.from_csr(indptr, indices), Xd, yd, m2, None, None, batch_size=B, sampling_hops=hops,
                             num_samples=K, optimizer_c=o2, e_cap=1 << 14, philox_seed=5, random_sampling=True)
    for it, tg in enumerate(batches):
        a = eager.step(tg, trace=True)
        b = graphed.step(tg)
        torch.cuda.synchronize()
        graphed.check()
        for hop in range(hops):
            ka = a["hops"][hop]["kept"]
            kc = int(b["kept_counts"][hop].item())
            assert kc == ka.numel() and torch.equal(b["kept"][hop][:kc], ka.to(torch.int32)), (it, hop)
        na = int(b["n_all"].item())
        assert torch.equal(b["all_nodes"][:na], a["all_nodes"])
        tol = dict(rtol=1e-5, atol=1e-6) if it == 0 else dict(rtol=2e-4, atol=3e-5)
        assert torch.allclose(b["logits"][:na], a["logits"], **tol), it
        assert abs(float(b["loss_c"]) - float(a["loss_c"])) <= 1e-5 * max(1.0, abs(float(a["loss_c"])))
        assert b["loss_gfn"] is None
        assert GraphedTrainer.edges_aggregated(b) == GrapesTrainer.edges_aggregated(a)
    assert graphed.graph_obj is not None
    for (k, p_), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.allclose(p_, q, rtol=1e-4, atol=1e-5), k
    with pytest.raises(ValueError):
        GraphedTrainer(DeviceGraph.from_csr(indptr, indices), Xd, yd, m2, None, None, batch_size=B)


@pytest.mark.parametrize("opt", ["no indicators", "hidden 64", "hidden 100", "hidden 50", "F 37", "F 602 hidden 64", "multilabel",
                                 "K above candidates", "reg_param", "dropout"])
def test_captured_step_option_matrix(opt):
    """The captured step against the eager step (which the other tests hold against the oracle) across the options that pick
    different kernels: indicator columns off, hidden widths with / without the bf16x3 and gate-bit forms (64: yes; 100: not a
    multiple of 32; 50: not a multiple of 4 either), odd and wide feature widths (transform-first order), BCE targets
    (main.py:120-123), and a k that exceeds the candidates (every neighbour kept, utils.py:44-51)."""
    _cuda()
    from grapes_amd import synth
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.step import GrapesTrainer
    from grapes_amd.step_graph import GraphedTrainer
    n, deg, F, C, B, K, hops, H = 20000, 9.0, 100, 6, 64, 48, 2, 256
    use_ind, multilabel = True, False
    if opt == "no indicators": use_ind = False
    elif opt.startswith("hidden"): H = int(opt.split()[1])
    elif opt == "F 37": F = 37
    elif opt == "F 602 hidden 64": F, H = 602, 64
    elif opt == "multilabel": multilabel = True
    elif opt == "K above candidates": K, B, deg = 4096, 16, 4.0
    reg = 0.05 if opt == "reg_param" else 0.0                       # main.py:260-261
    pdrop = 0.3 if opt == "dropout" else 0.0                        # main.py:110, modules/gcn.py:33,37
    indptr, indices = synth.synth_csr_numpy(n, deg, 300, seed=31)
    rng = np.random.default_rng(32)
    X = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).cuda()
    y = (torch.from_numpy((rng.random((n, C)) < 0.3).astype(np.float32)) if multilabel else torch.from_numpy(rng.integers(0, C, n))).cuda()
    batches = [torch.from_numpy(rng.permutation(n)[:B].astype(np.int64)).cuda() for _ in range(4)]
    ni = hops + 1 if use_ind else 0

    def build():
        torch.manual_seed(3)
        c, gf, z = GCN(F, [H, C], dropout=pdrop).cuda(), GCN(F + ni, [H, 1]).cuda(), GCN(F, [H, 1]).cuda()
        oc = torch.optim.Adam(c.parameters(), lr=1e-3, capturable=True)
        og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=1e-4, capturable=True)
        return c, gf, z, oc, og

    c, gf, z, oc, og = build()
    eager = GrapesTrainer(DeviceGraph.from_csr(indptr, indices), X, y, c, gf, z, sampling_hops=hops, num_samples=K, loss_coef=20.0,
                          use_indicators=use_ind, optimizer_c=oc, optimizer_gf=og, philox_seed=9, reg_param=reg)
    c2, gf2, z2, oc2, og2 = build()
    graphed = GraphedTrainer(DeviceGraph.from_csr(indptr, indices), X, y, c2, gf2, z2, batch_size=B, sampling_hops=hops,
                             num_samples=K, loss_coef=20.0, use_indicators=use_ind, optimizer_c=oc2, optimizer_gf=og2,
                             e_cap=1 << 15, philox_seed=9, reg_param=reg)
    if pdrop:    # the mask is the oracle generator's: kept iff philox_uniform(seed, offset, i) >= p, scaled by 1 / (1 - p)
        from grapes_amd import ops
        from oracle import portable_math as pm
        xd = torch.randn(301, 37, device="cuda")
        off_t = torch.tensor([123], dtype=torch.int64, device="cuda")
        yd, keep = ops.dropout_fwd(xd, pdrop, philox_seed=77, d_philox_offset=off_t)
        u = pm.philox_uniform(77, 123, xd.numel()).reshape(301, 37)
        kref = torch.from_numpy(u >= np.float32(pdrop)).cuda()
        assert torch.equal(keep.bool(), kref) and int(off_t) == 123 + (xd.numel() + 3) // 4
        assert torch.equal(yd, torch.where(kref, xd * (1.0 / (1.0 - pdrop)), torch.zeros_like(xd)))
        assert 0.25 < 1.0 - float(kref.float().mean()) < 0.35
        g_ = torch.randn_like(xd)
        assert torch.equal(ops.dropout_bwd(g_, keep, pdrop), torch.where(kref, g_ * (1.0 / (1.0 - pdrop)), torch.zeros_like(g_)))
    if reg:      # the regulariser itself against torch (unbiased variance over the classes, summed over the rows)
        from grapes_amd import ops
        lg = torch.randn(777, 13, device="cuda") * 3.0
        lgd = lg.double().requires_grad_(True)
        ref = reg * torch.sum(torch.var(lgd, dim=1)); ref.backward()
        assert abs(float(ops.logit_var_reg(lg, reg)) - float(ref.detach())) <= 1e-5 * float(ref.detach())
        dl0 = torch.randn_like(lg)
        got = ops.logit_var_reg(lg, reg, dlogits=dl0.clone())
        assert torch.allclose(got.double(), dl0.double() + lgd.grad, rtol=1e-5, atol=1e-6)
    for it, tg in enumerate(batches):
        a = eager.step(tg, trace=True)
        b = graphed.step(tg)
        torch.cuda.synchronize()
        graphed.check()
        for hop in range(hops):
            ka = a["hops"][hop]["kept"]
            kc = int(b["kept_counts"][hop].item())
            assert kc == ka.numel() and torch.equal(b["kept"][hop][:kc], ka.to(torch.int32)), (opt, it, hop)
        na = int(b["n_all"].item())
        assert torch.equal(b["all_nodes"][:na], a["all_nodes"])
        tol = dict(rtol=1e-5, atol=2e-6) if it == 0 else dict(rtol=3e-4, atol=5e-5)
        assert torch.allclose(b["logits"][:na], a["logits"], **tol), (opt, it)
        for key, t in (("loss_c", 1e-5), ("log_z", 1e-5), ("tot_log_prob", 2e-5), ("loss_gfn", 2e-4)):
            assert abs(float(b[key]) - float(a[key])) <= t * max(1.0, abs(float(a[key]))) * (1 if it == 0 else 20), (opt, it, key)
        assert GraphedTrainer.edges_aggregated(b) == GrapesTrainer.edges_aggregated(a)
    assert graphed.graph_obj is not None
    for m1, m2 in ((c, c2), (gf, gf2), (z, z2)):
        for (k, p_), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
            assert torch.allclose(p_, q, rtol=2e-4, atol=2e-5), (opt, k)


# ------------------------------------------------------------------------------ BASELINE-size properties
def test_products_scale_properties():
    """ogbn-products-shaped synthetic graph (N=2,449,029): size-independent properties of the hop
    pipeline — edge count = Σ degrees, compaction sorted/unique/complete, relabel round trip,
    top-k separation, sampled block ⊂ adjacency."""
    dev = _cuda()
    from grapes_amd import ops, synth
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.step import GrapesTrainer
    N, deg, maxdeg, F, C, B, K, hops = synth.CONFIGS["products"]
    rowptr, col = synth.synth_graph_device(N, deg, maxdeg, seed=0)
    dg = DeviceGraph(rowptr, col, N)
    assert abs(dg.nnz / N - deg) < 2.0
    degs = (rowptr[1:] - rowptr[:-1])
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    targets = torch.randperm(N, device="cuda", generator=gen)[:B]
    X = torch.randn(N, 8, device="cuda")
    tr = GrapesTrainer(dg, X, None, None, None, None, sampling_hops=hops, num_samples=K)
    inject = lambda hop, bn: torch.sin(bn.to(torch.float32) * 0.001 + hop)
    out = tr.step(targets, inject_logits_fn=inject, trace=True)
    prev = targets
    for hop in range(hops):
        h = out["hops"][hop]
        nb = h["neighborhoods"]
        assert nb.shape[1] == int(degs[prev.long()].sum())                         # every out-edge, once
        b = h["batch_nodes"].long()
        assert bool((b[1:] > b[:-1]).all())                                        # ascending & unique
        assert torch.equal(b, torch.unique(nb.reshape(-1).long()))
        nn_ = h["neighbor_nodes"].long()
        assert not bool(torch.isin(nn_, prev.long()).any())
        assert torch.equal(b[h["local_neighborhoods"].long()], nb.long())          # relabel round trip
        kept = h["kept"].long()
        assert kept.numel() == min(K, nn_.numel()) and bool(torch.isin(kept, nn_).all())
        khe = h["k_hop_edges"].long()
        nxt = torch.cat([targets, kept])
        assert bool(torch.isin(khe[0], nxt).all()) and bool(torch.isin(khe[1], prev.long()).all())
        # each sampled edge is an edge of A: binary search in the CSR row
        key = khe[0] * N + khe[1]
        if hop == 0:
            allk = torch.repeat_interleave(torch.arange(N, device=dev), degs) * N + col.long()
        pos = torch.searchsorted(allk, key)
        assert bool((allk[pos.clamp(max=allk.numel() - 1)] == key).all())
        prev = nxt
    assert int(dg.bits.ne(0).sum()) == 0 and int(dg.mult.ne(0).sum()) == 0


# ------------------------------------------------------------------ N2: losses + Adam on the device
@pytest.mark.parametrize("C,B,multi", [(47, 256, False), (7, 5, False), (172, 300, False), (41, 64, True)])
def test_classifier_loss_matches_torch(C, B, multi):
    """main.py:260,267: CrossEntropyLoss / BCEWithLogitsLoss over logits[local_target_ids] and its gradient."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(C + B)
    N, n_rows = 5000, B + 300
    logits = (rng.standard_normal((n_rows, C)) * 3).astype(np.float32)
    rows = rng.permutation(n_rows)[:B].astype(np.int32)
    tids = rng.permutation(N)[:B].astype(np.int32)
    y = (rng.random((N, C)) < 0.3).astype(np.float32) if multi else rng.integers(0, C, N)
    lt = torch.from_numpy(logits).requires_grad_(True)
    sel = lt[torch.from_numpy(rows).long()]
    tgt = torch.from_numpy(y)[torch.from_numpy(tids).long()]
    ref = (torch.nn.BCEWithLogitsLoss() if multi else torch.nn.CrossEntropyLoss())(sel, tgt)     # main.py:120-123
    ref.backward()
    loss, grad = ops.classifier_loss(_t(logits), _t(rows), _t(tids), _t(y))
    assert abs(float(loss) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref)))
    assert _close(grad.cpu().numpy(), lt.grad.numpy(), 1e-6)
    assert float(grad.cpu().abs().sum(1)[np.setdiff1d(np.arange(n_rows), rows)].max()) == 0.0


def test_gflownet_loss_matches_reference_formula():
    """main.py:272-282 (trajectory balance) and main.py:279 (REINFORCE)."""
    _cuda()
    from grapes_amd import ops
    stats = np.zeros((3, 6), np.float32); stats[:, 4] = [-812.25, -2400.5, -2399.75]
    lz_raw, lz_init, cost, coef = np.float32(3071.5), 1.25, np.float32(3.8125), 15227.124
    tot = np.float32(np.float32(stats[0, 4] + stats[1, 4]) + stats[2, 4])
    lz = np.float32(lz_raw - np.float32(lz_init))
    inner = np.float32(np.float32(lz + tot) + np.float32(np.float32(coef) * cost))
    o = ops.gflownet_loss(_t(stats), _t(np.array([cost])), coef, log_z_raw=_t(np.array([lz_raw])), log_z_init=lz_init).cpu().numpy()
    assert o[0] == np.float32(inner * inner) and o[1] == np.float32(2 * inner) and o[2] == lz and o[3] == tot
    o = ops.gflownet_loss(_t(stats), _t(np.array([cost])), coef, reinforce=True).cpu().numpy()
    assert o[0] == np.float32(-tot * cost) and o[1] == -cost


def test_fused_adam_matches_torch_adam():
    """main.py:268,289: one launch for both optimisers == torch.optim.Adam.step() on each, state included."""
    _cuda()
    from grapes_amd import ops
    torch.manual_seed(3)
    shapes = [(256, 104), (256,), (1, 256), (1,), (47, 256), (300, 7)]
    def make():
        torch.manual_seed(4)
        ps = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
        oa = torch.optim.Adam(ps[:4], lr=4.469e-4, capturable=True)
        ob = torch.optim.Adam(ps[4:], lr=2.556e-5, weight_decay=1e-3, betas=(0.8, 0.95), eps=1e-6, capturable=True)
        return ps, oa, ob
    pa, oa, ob = make()
    pb, oc, od = make()
    fused = None
    for it in range(7):
        gs = [torch.randn(s, device="cuda") * (10.0 ** (it - 3)) for s in shapes]
        for p, q, g in zip(pa, pb, gs):
            p.grad = g.clone()
            if q.grad is None:
                q.grad = g.clone()
            else:
                q.grad.copy_(g)
        oa.step(); ob.step()
        if fused is None:
            fused = ops.FusedAdam([oc, od])
        fused.step()
        for p, q in zip(pa, pb):
            assert torch.allclose(p, q, rtol=2e-6, atol=1e-7), it
    for o1, o2 in ((oa, oc), (ob, od)):
        for p, q in zip(o1.param_groups[0]["params"], o2.param_groups[0]["params"]):
            assert float(o1.state[p]["step"]) == float(o2.state[q]["step"]) == 7.0
            for key in ("exp_avg", "exp_avg_sq"):      # gradients span 6 decades: tolerance relative to the tensor's scale
                u, v = o1.state[p][key], o2.state[q][key]
                assert float((u - v).abs().max()) <= 2e-6 * float(u.abs().max()), key


@pytest.mark.parametrize("n,K,N", [(5000, 104, 256), (37, 100, 256), (33111, 104, 256), (4100, 8, 64), (2500, 64, 96), (9000, 124, 256)])
def test_w_stationary_gemm_is_bit_identical_to_tiled_gemm(n, K, N):
    """The W-stationary forward GEMM (all of W resident in LDS, 32-row panels streamed) walks k in the same order as
    the 128x128 tiled kernel => bit-identical outputs; both against an fp64 reference at 1e-5."""
    _cuda()
    from grapes_amd import _lib, ops
    lib = _lib.load_diag()      # (grapes_debug_*: measurement entry points of the diagnostic build; same kernels)
    rng = np.random.default_rng(n + K)
    x = _t(rng.standard_normal((n, K)).astype(np.float32)); w = _t((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    st = torch.cuda.current_stream().cuda_stream
    a = torch.empty(n, N, device="cuda"); b = torch.full((n, N), 7.0, device="cuda")
    _lib.check(lib.grapes_debug_gemm_fwd(x.data_ptr(), w.data_ptr(), a.data_ptr(), n, K, N, 0, st), "tiled")
    _lib.check(lib.grapes_debug_gemm_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), n, K, N, 16, st), "w-stationary")
    assert torch.equal(a, b)
    ref = (x.double() @ w.double().T)
    assert float((a.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    # through the product entry point (bias + ReLU epilogue, device-side row count with a larger capacity)
    bias = _t(rng.standard_normal(N).astype(np.float32))
    cap = n + 777
    xc = torch.cat([x, torch.full((777, K), float("nan"), device="cuda")])
    d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    out = ops.linear_bias_act_fwd(xc, w, bias, True, d_n=d_n)
    # (the product entry point may take the bf16x3 kernel — same accuracy, other rounding: test_split_bf16_gemm_*)
    want = torch.relu(ref + bias.double())
    assert out.shape == (cap, N)
    assert float((out[:n].double() - want).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("n,K,N,scale", [(37500, 104, 256, 1.0), (5000, 100, 256, 1e3), (4099, 128, 256, 1e-3), (2050, 64, 96, 1.0),
                                         (33, 8, 32, 1.0), (70001, 124, 224, 30.0)])
def test_split_bf16_gemm_is_as_accurate_as_the_fp32_mfma_gemm(n, K, N, scale):
    """The forward GEMM on the bf16 matrix pipe with every fp32 operand split exactly into three bf16 terms (six cross
    products, fp32 accumulation): its error against fp64, relative to sum |a.b| of each output, is no larger than the
    fp32-MFMA kernel's on the same data, and far inside the 1e-5 activation tolerance.  Shapes cover a partial last
    panel, K not a multiple of 16, N < 256 (idle wavefronts), one panel only."""
    _cuda()
    from grapes_amd import _lib
    lib = _lib.load_diag()      # (grapes_debug_*: measurement entry points of the diagnostic build; same kernels)
    rng = np.random.default_rng(n + K + N)
    x = _t((rng.standard_normal((n, K)) * scale).astype(np.float32)); w = _t((rng.standard_normal((N, K)) * 0.1).astype(np.float32))
    x[n // 2, :] = 0.0; x[n // 3, 0] = 1e30 * scale; w[0, :] = 0.0; w[1, 1] = -1e-30      # zeros, very large, very small
    st = torch.cuda.current_stream().cuda_stream
    ref = x.double() @ w.double().T
    mag = (x.double().abs() @ w.double().abs().T).clamp_min(1e-300)
    errs = {}
    for dbg in (0, 64):
        out = torch.full((n + 3, N), float("nan"), device="cuda")
        _lib.check(lib.grapes_debug_gemm_fwd(x.data_ptr(), w.data_ptr(), out.data_ptr(), n, K, N, dbg, st), "gemm")
        assert bool(torch.isnan(out[n:]).all()) and not bool(torch.isnan(out[:n]).any())       # rows beyond n untouched
        rel = (out[:n].double() - ref).abs() / mag
        errs[dbg] = (float(rel.max()), float((rel ** 2).mean().sqrt()))
    assert errs[64][0] <= 1.05 * errs[0][0] + 1e-9 and errs[64][1] <= 1.05 * errs[0][1] + 1e-10, errs
    assert errs[64][0] < 1e-6


@pytest.mark.parametrize("n,fi,fo", [(5000, 104, 256), (777, 100, 64), (40000, 104, 256)])
def test_gated_dw_gemm_with_rank1_operand_and_head_gradient(n, fi, fo):
    """Backward of  first layer -> ReLU -> 1-wide head  in ONE split-K GEMM (step_graph._head_bwd): with
    dAct = dh2 ⊗ w2 formed on load and masked by the ReLU output,  dW1 = (dAct ⊙ [act>0])ᵀ ax,  db1 = its column sums,
    and the head's own  dW2 = dh2ᵀ act  gathered from the same gate tiles; against fp64, incl. accumulation and a
    device-side row count below the capacity."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(n + fo)
    cap = n + 333
    ax = _t(rng.standard_normal((cap, fi)).astype(np.float32))
    act = torch.relu(_t(rng.standard_normal((cap, fo)).astype(np.float32)))
    dh2 = _t((rng.standard_normal(cap) * 0.1).astype(np.float32))
    w2 = _t(rng.standard_normal(fo).astype(np.float32))
    d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    dw = torch.full((fo, fi), 0.5, device="cuda"); db = torch.full((fo,), -1.0, device="cuda"); dwh = torch.full((fo,), 2.0, device="cuda")
    ops.linear_bwd_weight_gated(None, ax, gate=act, d_n=d_n, dw=dw, dbias=db, accumulate=True, row_scale=dh2, col_vec=w2,
                                dw_head=dwh)
    A, G, X = dh2[:n].double()[:, None] * w2.double()[None, :], act[:n].double(), ax[:n].double()
    Ag = A * (G > 0)
    ref_dw, ref_db, ref_h = 0.5 + Ag.T @ X, -1.0 + Ag.sum(0), 2.0 + dh2[:n].double() @ G
    for got, ref in ((dw, ref_dw), (db, ref_db), (dwh, ref_h)):
        assert float((got.double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    dw2 = torch.empty_like(dw); db2 = torch.empty_like(db); dwh2 = torch.empty_like(dwh)
    ops.linear_bwd_weight_gated(None, ax, gate=act, d_n=d_n, dw=dw2, dbias=db2, accumulate=False, row_scale=dh2, col_vec=w2,
                                dw_head=dwh2)
    assert float((dwh2.double() - (ref_h - 2.0)).abs().max()) <= 2e-5 * max(1.0, float(ref_h.abs().max()))
    assert float((dw2.double() - (ref_dw - 0.5)).abs().max()) <= 2e-5 * max(1.0, float(ref_dw.abs().max()))


@pytest.mark.parametrize("n,K,N", [(1025, 256, 256), (1025, 256, 47), (1025, 100, 256), (37, 47, 256), (4096, 64, 96), (300, 13, 7)])
def test_one_shot_gemm_for_few_rows_against_tiled_gemm(n, K, N):
    """The few-row GEMM (whole K extent of both operands in LDS after ONE round trip; K summed as four quarters in a fixed
    order) against the tiled kernel (same products, another summation tree: a few ulps) and, through linear_fwd /
    linear_bwd_input (device-side row count), against fp64; and it is deterministic."""
    _cuda()
    from grapes_amd import _lib, ops
    lib = _lib.load_diag()      # (grapes_debug_*: measurement entry points of the diagnostic build; same kernels)
    rng = np.random.default_rng(n + K + N)
    x = _t(rng.standard_normal((n, K)).astype(np.float32)); w = _t((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    st = torch.cuda.current_stream().cuda_stream
    a = torch.empty(n, N, device="cuda"); b = torch.full((n, N), 7.0, device="cuda")
    _lib.check(lib.grapes_debug_gemm_fwd(x.data_ptr(), w.data_ptr(), a.data_ptr(), n, K, N, 0, st), "tiled")
    _lib.check(lib.grapes_debug_gemm_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), n, K, N, 32, st), "one-shot")
    assert float((a - b).abs().max()) <= 4e-6 * max(1.0, float(a.abs().max()))
    b2 = torch.empty_like(b)
    _lib.check(lib.grapes_debug_gemm_fwd(x.data_ptr(), w.data_ptr(), b2.data_ptr(), n, K, N, 32, st), "one-shot")
    assert torch.equal(b, b2)
    cap = n + 50
    xc = torch.cat([x, torch.full((50, K), float("nan"), device="cuda")])
    d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    h = ops.linear_fwd(xc, w, d_n=d_n)
    ref = x.double() @ w.double().T
    assert h.shape == (cap, N) and float((h[:n].double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    dh = torch.cat([_t(rng.standard_normal((n, N)).astype(np.float32)), torch.full((50, N), float("nan"), device="cuda")])
    dx = ops.linear_bwd_input(dh, w, d_n=d_n)                                   # dX = dH · W   (K = N here: k-major B)
    ref = dh[:n].double() @ w.double()
    assert dx.shape == (cap, K) and float((dx[:n].double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))


def test_fused_expand_matches_two_launch_expand():
    """get_neighborhoods in one launch (every workgroup rebuilds the row-length scan) == offsets + expand, including a
    hub row, empty rows, a device-side query count and the overflow flag; and == the oracle (utils.py:74-82)."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(77)
    n = 20000
    ei = rng.integers(0, n, (2, 150000)); ei[0, :9000] = 5
    indptr, indices = O.build_csr(np.concatenate([ei, ei[::-1]], axis=1), n)
    indptr[-1:]  # noqa: B018
    rowptr, col = _t(indptr), _t(indices, torch.int32)
    for m, live in ((512, 512), (512, 300), (1, 1), (2048, 2000), (700, 700)):
        nodes = rng.permutation(n)[:m].astype(np.int32); nodes[0] = 5
        iso = np.setdiff1d(np.arange(n), np.unique(ei))[:1]
        if iso.size and m > 2:
            nodes[2] = iso[0]                                          # an empty row
        d_m = torch.tensor([live], dtype=torch.int32, device="cuda")
        ref = O.get_neighborhoods(nodes[:live].astype(np.int64), indptr, indices)
        e = ref.shape[1]
        st = torch.zeros(1, dtype=torch.int32, device="cuda")
        src, dst, d_e, eoff = ops.frontier_expand_fused(rowptr, col, _t(nodes), e + 100, d_m=d_m, status=st)
        eoff2, d_e2 = ops.frontier_offsets(rowptr, _t(nodes), d_m=d_m)
        assert int(d_e) == e == int(d_e2) and torch.equal(eoff[:live + 1], eoff2[:live + 1]) and int(st) == 0
        assert np.array_equal(src[:e].cpu().numpy().astype(np.int64), ref[0]) and np.array_equal(dst[:e].cpu().numpy().astype(np.int64), ref[1])
        src, dst, d_e, _ = ops.frontier_expand_fused(rowptr, col, _t(nodes), max(e // 2, 1), d_m=d_m, status=st)   # too small
        assert int(st) & 1 and int(d_e) == e
    # the hop's bitmap marks in the same launch == bitmap_mark_hop
    W = (n + 63) // 64
    nodes = rng.permutation(n)[:600].astype(np.int32); nodes[0] = 5
    d_m = torch.tensor([555], dtype=torch.int32, device="cuda")
    pb, bb = torch.zeros(W, dtype=torch.int64, device="cuda"), torch.zeros(W, dtype=torch.int64, device="cuda")
    src, dst, d_e, eoff = ops.frontier_expand_fused(rowptr, col, _t(nodes), 1 << 17, d_m=d_m, status=st, mark_prev_bits=pb,
                                                    mark_bits=bb, num_nodes=n)
    pb2, bb2 = torch.zeros_like(pb), torch.zeros_like(bb)
    ops.bitmap_mark_hop(pb2, bb2, None, _t(nodes), eoff, dst, n, d_m=d_m, d_e=d_e, status=st)
    assert torch.equal(pb, pb2) and torch.equal(bb, bb2) and int(pb.ne(0).sum()) > 0
    # ... and the slice re-mark (un-mark / mark / clear another bitmap's words) in the same launch == slice_remark
    mult = torch.zeros(n, dtype=torch.int32, device="cuda"); mult2 = torch.zeros_like(mult)
    un, mk = _t(np.arange(100, 400), torch.int32), _t(np.arange(5000, 5600), torch.int32)
    mult[un.long()] = 1; mult2[un.long()] = 1
    other = torch.full((W,), -1, dtype=torch.int64, device="cuda"); other2 = other.clone()
    cnt = lambda v: torch.tensor([v], dtype=torch.int32, device="cuda")
    pb.zero_(); bb.zero_()
    ops.frontier_expand_fused(rowptr, col, _t(nodes), 1 << 17, d_m=d_m, status=st, mark_prev_bits=pb, mark_bits=bb, num_nodes=n,
                              remark=dict(mult=mult, unmark=(un, cnt(250)), mark=(mk, None), clear=(_t(nodes), d_m), clear_bits=other))
    ops.slice_remark(mult2, unmark=(un, cnt(250)), mark=(mk, None), clear=(_t(nodes), d_m), clear_bits=other2)
    assert torch.equal(mult, mult2) and torch.equal(other, other2) and torch.equal(pb, pb2) and torch.equal(bb, bb2)
    # a device-side count of zero: no edges, offsets [0]
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    src, dst, d_e, eoff = ops.frontier_expand_fused(rowptr, col, _t(nodes), 64, d_m=torch.zeros(1, dtype=torch.int32, device="cuda"),
                                                    status=st)
    assert int(d_e) == 0 and int(eoff[0]) == 0 and int(st) == 0


def test_multi_hop_gated_dw_equals_sum_of_single_hop_launches():
    """One split-K GEMM for the three hops that share the sampler GCN's weights == the three accumulating launches
    (same operands, device-side row counts below capacity), and both against fp64."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(5)
    fi, fo = 104, 256
    ns, caps = [9000, 21000, 300], [12000, 21000, 4000]
    gates, xs, rss, dns = [], [], [], []
    ref_dw = torch.zeros(fo, fi, dtype=torch.float64, device="cuda"); ref_db = torch.zeros(fo, dtype=torch.float64, device="cuda")
    ref_h = torch.zeros(fo, dtype=torch.float64, device="cuda")
    w2 = _t(rng.standard_normal(fo).astype(np.float32))
    for n, cap in zip(ns, caps):
        g = torch.relu(_t(rng.standard_normal((cap, fo)).astype(np.float32)))
        x = _t(rng.standard_normal((cap, fi)).astype(np.float32))
        rs = _t((rng.standard_normal(cap) * 0.1).astype(np.float32))
        g[n:] = float("nan"); x[n:] = float("nan")                      # rows beyond the live count must never be read
        gates.append(g); xs.append(x); rss.append(rs); dns.append(torch.tensor([n], dtype=torch.int32, device="cuda"))
        A = (rs[:n].double()[:, None] * w2.double()[None, :]) * (g[:n] > 0)
        ref_dw += A.T @ x[:n].double(); ref_db += A.sum(0); ref_h += rs[:n].double() @ g[:n].double()
    dw = torch.empty(fo, fi, device="cuda"); db = torch.empty(fo, device="cuda"); dh = torch.empty(fo, device="cuda")
    ops.linear_bwd_weight_gated_multi(gates, xs, rss, dns, w2, dw, dbias=db, dw_head=dh, accumulate=False)
    for got, ref in ((dw, ref_dw), (db, ref_db), (dh, ref_h)):
        assert float((got.double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    dw1 = torch.empty_like(dw); db1 = torch.empty_like(db); dh1 = torch.empty_like(dh)
    for h in range(3):
        ops.linear_bwd_weight_gated(None, xs[h], gate=gates[h], d_n=dns[h], dw=dw1, dbias=db1, accumulate=h > 0, row_scale=rss[h],
                                    col_vec=w2, dw_head=dh1)
    for a, b in ((dw, dw1), (db, db1), (dh, dh1)):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
    # accumulate on top of existing gradients
    dw2 = torch.full_like(dw, 3.0)
    ops.linear_bwd_weight_gated_multi(gates[:2], xs[:2], rss[:2], dns[:2], w2, dw2, accumulate=True)
    part = torch.empty_like(dw)
    ops.linear_bwd_weight_gated_multi(gates[:2], xs[:2], rss[:2], dns[:2], w2, part, accumulate=False)
    assert float((dw2 - 3.0 - part).abs().max()) <= 1e-5 * max(1.0, float(part.abs().max()))


def test_small_graph_batch_prepare_equals_single_prepares():
    """Three small graphs over the same node set in ONE launch == three single launches (the classifier's per-layer
    subgraphs, main.py:252-256): CSRs, dinv, work items, row heads."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(41)
    n, N = 900, 50000
    ids = np.sort(rng.permutation(N)[:n]).astype(np.int32)
    node_map = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    node_map[_t(ids).long()] = torch.arange(n, dtype=torch.int32, device="cuda")
    lists, st = [], torch.zeros(1, dtype=torch.int32, device="cuda")
    for e in (700, 3000, 40):
        s_loc = np.sort(rng.integers(0, n, e)); d_loc = rng.integers(0, n, e)
        d_loc[:50] = 3                                                     # a target with many in-edges
        order = np.lexsort((d_loc, s_loc)); s_loc, d_loc = s_loc[order], d_loc[order]
        cap = e + 100
        pad = lambda v: _t(np.concatenate([ids[v], np.full(cap - e, ids[0])]), torch.int32)
        lists.append((pad(s_loc), pad(d_loc), torch.tensor([e], dtype=torch.int32, device="cuda")))
    d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    batch = ops.PreparedGraph.small_batch(lists, n + 20, d_n=d_n, status=st, node_map=node_map, head_ids=_t(np.concatenate([ids, np.zeros(20, np.int32)]), torch.int32))
    assert int(st) == 0 and len(batch) == 3
    for (es, ed, d_e), got in zip(lists, batch):
        hid = got.head_ids
        ref = ops.PreparedGraph(es, ed, n + 20, d_n=d_n, d_e=d_e, status=st, src_grouped=True, node_map=node_map, head_ids=hid)
        ne = int(ref.rowptr_t[n])
        assert torch.equal(ref.rowptr_t[:n + 1], got.rowptr_t[:n + 1]) and torch.equal(ref.rowptr_s[:n + 1], got.rowptr_s[:n + 1])
        assert torch.equal(ref.csr_src[:ne], got.csr_src[:ne]) and torch.equal(ref.csr_dst[:ne], got.csr_dst[:ne])
        assert torch.equal(ref.dinv[:n], got.dinv[:n]) and ref.n_long.tolist()[:3] == got.n_long.tolist()[:3]
        assert torch.equal(ref.row_head[:n], got.row_head[:n])
    assert int(st) == 0


def test_step_losses_one_launch_equals_the_separate_launches():
    """grapes_step_losses == tensormap lookup + classifier_loss + reduce_sum(mean) + gflownet_loss, bit for bit."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(77)
    N, n_rows, C, B, hops = 5000, 700, 47, 256, 3
    targets = _t(rng.permutation(N)[:B], torch.int32)
    node_map = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    node_map[targets.long()] = _t(rng.permutation(n_rows)[:B], torch.int32)
    logits = _t(rng.standard_normal((n_rows, C)).astype(np.float32))
    y = _t(rng.integers(0, C, N), torch.int64)
    stats = _t(rng.standard_normal((hops, 6)).astype(np.float32))
    zcap, nz = 3000, 2111
    zout = _t(rng.standard_normal(zcap).astype(np.float32)); zout[nz:] = float("nan")
    d_nz = torch.tensor([nz], dtype=torch.int32, device="cuda")
    for reinforce in (False, True):
        loss, dl, out4 = ops.step_losses(logits, node_map, targets, y, stats, 1e4, z_out=zout, d_nz=d_nz, log_z_init=7.0,
                                         reinforce=reinforce)
        rows = ops.tensormap_map(node_map, targets)
        loss_r, dl_r = ops.classifier_loss(logits, rows, targets, y)
        zraw = ops.reduce_sum(zout, mean=True, d_n=d_nz)
        out_r = ops.gflownet_loss(stats, loss_r, 1e4, log_z_raw=zraw, log_z_init=7.0, reinforce=reinforce)
        assert torch.equal(loss, loss_r) and torch.equal(dl, dl_r) and torch.equal(out4, out_r)
        loss1, dl1, out41 = ops.step_losses(logits, node_map, targets, y, stats, 1e4, z_out=zout, d_nz=d_nz, log_z_init=7.0,
                                            reinforce=reinforce, many_workgroups=False)     # the single-workgroup form
        assert torch.equal(loss1, loss_r) and torch.equal(dl1, dl_r) and torch.equal(out41, out_r)
    # no log-Z head (random sampling): log_z = 0
    _, _, out4 = ops.step_losses(logits, node_map, targets, y, stats, 2.0)
    assert torch.equal(out4, ops.gflownet_loss(stats, loss_r, 2.0))
    # multilabel targets, more classes than a wavefront, a batch that is not a multiple of anything
    C2, B2 = 121, 301
    t2 = _t(rng.permutation(N)[:B2], torch.int32)
    nm2 = torch.full((N,), -1, dtype=torch.int32, device="cuda"); nm2[t2.long()] = _t(rng.permutation(n_rows)[:B2], torch.int32)
    lg2 = _t(rng.standard_normal((n_rows, C2)).astype(np.float32)); y2 = _t((rng.random((N, C2)) < 0.3).astype(np.float32))
    for kw in (dict(), dict(many_workgroups=False)):
        l2, d2, o2 = ops.step_losses(lg2, nm2, t2, y2, stats, 3.0, z_out=zout, d_nz=d_nz, **kw)
        lr, dr = ops.classifier_loss(lg2, ops.tensormap_map(nm2, t2), t2, y2)
        assert torch.equal(l2, lr) and torch.equal(d2, dr)
        assert torch.equal(o2, ops.gflownet_loss(stats, lr, 3.0, log_z_raw=ops.reduce_sum(zout, mean=True, d_n=d_nz)))
    assert int(ops._ticket(torch.device("cuda", 0)).ne(0).sum()) == 0


def test_slice_remark_and_epoch_advance():
    """The between-hops form of slice marking (un-mark old samples + mark new ones + clear bitmap words in one launch),
    the final un-mark inside bitmap_mark_lists, and indicator_mark that advances the device epoch itself."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(3)
    N = 100000
    perm = rng.permutation(N)
    targets, k0, k1 = (_t(np.sort(perm[a:b]), torch.int32) for a, b in ((0, 300), (300, 900), (900, 1400)))
    cap = lambda t, extra: torch.cat([t, torch.zeros(extra, dtype=torch.int32, device="cuda")])
    mult = torch.zeros(N, dtype=torch.int32, device="cuda")
    bits = torch.full(((N + 63) // 64,), -1, dtype=torch.int64, device="cuda")
    cnt = lambda t: torch.tensor([t.numel()], dtype=torch.int32, device="cuda")
    ops.slice_remark(mult, mark=(targets, None), clear=(targets, None), clear_bits=bits)
    ref = np.zeros(N, np.int32); ref[targets.cpu().numpy()] = 1
    assert np.array_equal(mult.cpu().numpy(), ref)
    words = np.unique(targets.cpu().numpy() >> 6)
    refbits = np.full(bits.numel(), -1, np.int64); refbits[words] = 0
    assert np.array_equal(bits.cpu().numpy(), refbits)
    ops.slice_remark(mult, mark=(cap(k0, 50), cnt(k0)))                    # device-side counts below capacity
    ref[k0.cpu().numpy()] = 1
    assert np.array_equal(mult.cpu().numpy(), ref)
    ops.slice_remark(mult, unmark=(cap(k0, 7), cnt(k0)), mark=(cap(k1, 9), cnt(k1)))
    ref[k0.cpu().numpy()] = 0; ref[k1.cpu().numpy()] = 1
    assert np.array_equal(mult.cpu().numpy(), ref)
    out_bits = torch.zeros_like(bits)
    ops.bitmap_mark_lists(out_bits, None, [(targets, None), (cap(k0, 7), cnt(k0)), (cap(k1, 9), cnt(k1))], N, unmark_mult=mult)
    assert int(mult.ne(0).sum()) == 0
    got = np.unpackbits(out_bits.cpu().numpy().view(np.uint8), bitorder="little")[:N]
    allids = np.zeros(N, np.uint8); allids[np.concatenate([t.cpu().numpy() for t in (targets, k0, k1)])] = 1
    assert np.array_equal(got, allids)
    # epoch advance: marks carry *d_epoch + 1 and the counter is stored back
    code = torch.zeros(N, dtype=torch.int32, device="cuda")
    ep = torch.tensor([41], dtype=torch.int32, device="cuda")
    ops.indicator_mark(code, targets, 0, 2, d_epoch=ep, advance_epoch=True)
    assert int(ep) == 42
    assert bool((code[targets.long()] == ((42 << 8) | 4)).all()) and int(code.ne(0).sum()) == targets.numel()
    ops.indicator_mark(code, k0, 0, 0, d_epoch=ep)
    assert int(ep) == 42 and bool((code[k0.long()] == ((42 << 8) | 1)).all())


@pytest.mark.parametrize("N", [4000, 2_449_029, 9_000_000, 20_000_000, 111_059_956])   # last: ogbn-papers100M
def test_one_launch_compaction_equals_two_launch_compaction(N):
    """The look-back form of frontier_compact (workgroup totals through the `sync` scratch) == the counting + emitting
    form, over repeated launches that share the scratch; the scratch is zero again after every launch.  The largest N
    needs more workgroups than the scratch has slots and must take the two-launch form by itself."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(N)
    W = (N + 63) // 64
    sync = ops.sync_scratch("cuda")
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    for rep in range(6):
        m = [5, 3000, 200000, 1, 40000, 777][rep]
        ids = _t(rng.integers(0, N, m), torch.int32)
        prev = ids[: m // 3].contiguous()
        outs = []
        for one in (True, False):
            bits = torch.zeros(W, dtype=torch.int64, device="cuda"); pbits = torch.zeros(W, dtype=torch.int64, device="cuda")
            node_map = torch.full((N,), -1, dtype=torch.int32, device="cuda")
            ops.bitmap_mark(bits, None, ids, N)
            if prev.numel():
                ops.bitmap_mark(pbits, None, prev, N)
            b, nb, nbl, c = ops.frontier_compact(bits, None, pbits, N, m + 8, node_map=node_map, status=st, one_launch=one)
            assert int(bits.ne(0).sum()) == 0
            outs.append((b, nb, nbl, c, node_map))
            assert int(sync.ne(0).sum()) == 0
        (b1, n1, l1, c1, m1), (b2, n2, l2, c2, m2) = outs
        nb_, nn_ = c2.tolist()
        assert c1.tolist() == [nb_, nn_] and nb_ == len(np.unique(ids.cpu().numpy()))
        assert torch.equal(b1[:nb_], b2[:nb_]) and torch.equal(n1[:nn_], n2[:nn_]) and torch.equal(l1[:nn_], l2[:nn_])
        assert torch.equal(m1, m2)
    assert int(st) == 0


def test_fp32_mfma_path_of_the_aggregate_first_gemms_still_works():
    """GRAPES_GEMM_SPLIT=0 (read once per process, hence a child process) routes the forward and the gated dW GEMM of
    the aggregate-first layers to the fp32-MFMA kernels; both paths agree with fp64 and with each other at 1e-5."""
    _cuda()
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, sys, numpy as np, torch
        sys.path.insert(0, os.getcwd())
        from grapes_amd import ops
        rng = np.random.default_rng(11)
        n, cap, fi, fo = 9000, 9500, 104, 256
        t = lambda a: torch.from_numpy(a).cuda()
        x = t(rng.standard_normal((cap, fi)).astype(np.float32)); w = t((rng.standard_normal((fo, fi)) * 0.1).astype(np.float32))
        b = t(rng.standard_normal(fo).astype(np.float32)); w2 = t(rng.standard_normal(fo).astype(np.float32))
        rs = t((rng.standard_normal(cap) * 0.1).astype(np.float32))
        d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
        out = ops.linear_bias_act_fwd(x, w, b, True, d_n=d_n)
        ref = torch.relu(x[:n].double() @ w.double().T + b.double())
        e1 = float((out[:n].double() - ref).abs().max() / ref.abs().max())
        dw = torch.empty(fo, fi, device="cuda"); db = torch.empty(fo, device="cuda"); dh = torch.empty(fo, device="cuda")
        gate = torch.zeros(cap, fo, device="cuda")                  # the SAME mask for both kernels: the ReLU threshold is
        gate[:n] = torch.where(ref > 1e-3, ref, torch.zeros_like(ref)).float()   # a discontinuity, keep clear of it
        ops.linear_bwd_weight_gated(None, x, gate=gate, d_n=d_n, dw=dw, dbias=db, accumulate=False, row_scale=rs, col_vec=w2, dw_head=dh)
        A = (rs[:n].double()[:, None] * w2.double()[None, :]) * (gate[:n] > 0)
        rdw, rdb, rdh = A.T @ x[:n].double(), A.sum(0), rs[:n].double() @ gate[:n].double()
        e2 = max(float((g.double() - r).abs().max() / r.abs().max()) for g, r in ((dw, rdw), (db, rdb), (dh, rdh)))
        print("ERR", e1, e2)
        torch.save({"out": out[:n].cpu(), "dw": dw.cpu()}, sys.argv[1])
    """)
    import tempfile
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for flag in ("1", "0"):
            env = dict(os.environ, GRAPES_GEMM_SPLIT=flag, GRAPES_DIAG="1")     # (a switch of the diagnostic build)
            path = os.path.join(td, f"o{flag}.pt")
            r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=300,
                               cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            assert r.returncode == 0, r.stderr[-2000:]
            e1, e2 = (float(v) for v in r.stdout.split("ERR")[1].split()[:2])
            assert e1 <= 1e-5 and e2 <= 2e-5, (flag, e1, e2)
            res[flag] = torch.load(path)
    for k in ("out", "dw"):
        a, b = res["1"][k].double(), res["0"][k].double()
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
    assert not torch.equal(res["1"]["out"], res["0"]["out"])          # really two different kernels


@pytest.mark.parametrize("e,dup", [(7, 1), (5000, 1), (300000, 3), (1_000_000, 1)])
def test_one_launch_slice_filter_equals_two_launch_filter(e, dup):
    """slice_filter through the look-back scratch (4 edges per thread) == the counting + emitting form: same survivors,
    same order, multiplicities honoured, device-side edge count below capacity, overflow flagged the same way."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(e)
    N = 50000
    mult = torch.zeros(N, dtype=torch.int32, device="cuda")
    marked = _t(rng.permutation(N)[:3000], torch.int64)
    mult[marked] = _t(rng.integers(1, dup + 1, 3000), torch.int32)
    src = _t(rng.integers(0, N, e + 100), torch.int32); dst = _t(rng.integers(0, N, e + 100), torch.int32)
    d_e = torch.tensor([e], dtype=torch.int32, device="cuda")
    for cap in (e * dup + 10, 50):
        outs = []
        for one in (True, False):
            st = torch.zeros(1, dtype=torch.int32, device="cuda")
            a, b, c = ops.slice_filter(mult, src, dst, cap, d_e=d_e, status=st, one_launch=one)
            outs.append((a, b, int(c), int(st)))
            assert int(ops.sync_scratch("cuda").ne(0).sum()) == 0
        (a1, b1, c1, s1), (a2, b2, c2, s2) = outs
        assert c1 == c2 and s1 == s2 and torch.equal(a1[:c1], a2[:c1]) and torch.equal(b1[:c1], b2[:c1])
    keep = mult[dst[:e].long()].cpu().numpy()
    ref_dst = np.repeat(dst[:e].cpu().numpy(), keep)
    a, b, c = ops.slice_filter(mult, src, dst, e * dup + 10, d_e=d_e)
    assert int(c) == len(ref_dst) and np.array_equal(b[:int(c)].cpu().numpy(), ref_dst)


@pytest.mark.parametrize("n,cap,K,N", [(37501, 37600, 104, 256), (5000, 5000, 100, 256), (2100, 4100, 64, 96), (700, 800, 104, 256), (20, 3000, 104, 256), (64, 2048, 8, 32)])
def test_forward_gemm_with_fused_head_projection(n, cap, K, N):
    """out = ReLU(x Wᵀ + b) and head = out · w2 from the same launch (summed from the output tiles in registers): the
    activations equal the plain entry point's bit for bit, the head equals out @ w2 in fp64 at 1e-5; rows beyond the
    device-side count are not written.  Covers a partial last panel, idle wavefronts (N < 256) and the two-launch
    fallback for few rows."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(n + N)
    x = _t(rng.standard_normal((cap, K)).astype(np.float32)); x[n:] = float("nan")
    w = _t((rng.standard_normal((N, K)) * 0.1).astype(np.float32)); b = _t(rng.standard_normal(N).astype(np.float32))
    w2 = _t(rng.standard_normal((1, N)).astype(np.float32))
    d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    out, head = ops.linear_bias_act_head_fwd(x, w, b, True, w2, d_n=d_n)
    ref_out = ops.linear_bias_act_fwd(x, w, b, True, d_n=d_n)
    assert torch.equal(out[:n], ref_out[:n])
    ref_head = out[:n].double() @ w2.double().T
    assert float((head[:n].double() - ref_head).abs().max()) <= 1e-5 * max(1.0, float(ref_head.abs().max()))
    # determinism: the same launch twice gives the same bits
    out2, head2 = ops.linear_bias_act_head_fwd(x, w, b, True, w2, d_n=d_n)
    assert torch.equal(head[:n], head2[:n])


def test_strided_input_forms_of_the_split_gemms():
    """The forward (+ head) and the rank-1 gated dW GEMM reading the leading 100 columns of a 104-wide matrix in place
    (the log-Z net at hop 0 reads the sampler net's aggregated input) == the same calls on a contiguous copy, bit for
    bit; rows beyond the device-side count (NaN) are never read."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(21)
    n, cap, fw, fi, fo = 12701, 13000, 104, 100, 256
    assert ops.split_gemm_available(cap, fi, fo)
    wide = _t(rng.standard_normal((cap, fw)).astype(np.float32)); wide[n:] = float("nan")
    x_view = wide[:, :fi]; x_copy = x_view.contiguous()
    w = _t((rng.standard_normal((fo, fi)) * 0.1).astype(np.float32)); b = _t(rng.standard_normal(fo).astype(np.float32))
    w2 = _t(rng.standard_normal((1, fo)).astype(np.float32)); rs = _t((rng.standard_normal(cap) * 0.1).astype(np.float32))
    d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    o1, h1 = ops.linear_bias_act_head_fwd_strided(x_view, w, b, True, w2, d_n=d_n)
    o2, h2 = ops.linear_bias_act_head_fwd(x_copy, w, b, True, w2, d_n=d_n)
    assert torch.equal(o1[:n], o2[:n]) and torch.equal(h1[:n], h2[:n])
    g = [torch.empty(fo, fi, device="cuda") for _ in range(2)]; db = [torch.empty(fo, device="cuda") for _ in range(2)]
    dh = [torch.empty(fo, device="cuda") for _ in range(2)]
    ops.linear_bwd_weight_gated_strided(x_view, o1, rs, w2.view(-1), g[0], dbias=db[0], dw_head=dh[0], d_n=d_n)
    ops.linear_bwd_weight_gated(None, x_copy, gate=o2, d_n=d_n, dw=g[1], dbias=db[1], accumulate=False, row_scale=rs,
                                col_vec=w2.view(-1), dw_head=dh[1])
    assert torch.equal(g[0], g[1]) and torch.equal(db[0], db[1]) and torch.equal(dh[0], dh[1])
    assert not bool(torch.isnan(g[0]).any())
    from grapes_amd import _lib
    with pytest.raises(_lib.GrapesHipError):                      # few rows: no bf16x3 kernel, so no strided form
        ops.linear_bias_act_head_fwd_strided(wide[:100, :fi], w, b, True, w2)


def test_sampler_head_backward_of_all_hops_in_two_launches():
    """sampler_head_bwd_multi == per hop [zero fill, bernoulli_logprob_bwd scattered through nb_local, by-source narrow
    aggregation], and its sum_out == the sum of all hops' d logits (fp64 check); cand_pos from the compaction is the
    inverse of nb_local."""
    _cuda()
    from grapes_amd import ops
    from grapes_amd.graph import DeviceGraph
    rng = np.random.default_rng(5)
    N = 60000
    ei = rng.integers(0, N, (2, N * 8))
    indptr, indices = O.build_csr(np.concatenate([ei, ei[::-1]], axis=1), N)
    dg = DeviceGraph.from_csr(indptr, indices)
    st = dg.status
    hops = []
    for m in (200, 450, 90):
        prev = _t(rng.permutation(N)[:m], torch.int32)
        src, dst, d_e, eoff = ops.frontier_expand_fused(dg.rowptr, dg.col, prev, 1 << 16, status=st)
        ops.bitmap_mark_hop(dg.prev_bits, dg.bits, None, prev, eoff, dst, N, d_e=d_e, status=st)
        n_cap = 40000
        batch, neigh, nbl, counts, cand_pos = ops.frontier_compact(dg.bits, None, dg.prev_bits, N, n_cap, node_map=dg.node_map,
                                                                   status=st, want_cand_pos=True)
        ops.bitmap_clear(dg.prev_bits, prev)
        nb, nn = counts.tolist()
        inv = torch.full((nb,), -1, dtype=torch.int32, device="cuda")
        inv[nbl[:nn].long()] = torch.arange(nn, dtype=torch.int32, device="cuda")
        assert torch.equal(cand_pos[:nb], inv)
        prep = ops.PreparedGraph(src, dst, n_cap, d_n=counts[0:1], d_e=d_e, status=st, src_grouped=True, items_fwd=False,
                                 node_map=dg.node_map)
        logit = _t(rng.standard_normal(n_cap).astype(np.float32))
        mask = _t((rng.random(n_cap) < 0.2).astype(np.float32))
        hops.append(dict(logit=logit, mask=mask, cand_pos=cand_pos, prep=prep, nbl=nbl, d_nn=counts[1:2], nb=nb, nn=nn))
    assert int(st) == 0
    scale = torch.tensor([-1.7], device="cuda")
    tot = torch.full((1,), 3.0, device="cuda")
    dlog, dh = ops.sampler_head_bwd_multi([h["logit"] for h in hops], [h["mask"] for h in hops], [h["cand_pos"] for h in hops],
                                          [h["prep"] for h in hops], d_grad_scale=scale, sum_out=tot, accumulate_sum=True)
    ref_tot = 3.0
    for q, h in enumerate(hops):
        ref = torch.zeros_like(h["logit"])
        ops.bernoulli_logprob_bwd(h["logit"], h["mask"], d_grad_scale=scale, logit_index=h["nbl"], out=ref, d_n=h["d_nn"])
        assert torch.equal(dlog[q][:h["nb"]], ref[:h["nb"]])
        rdh, _ = ops.gcn_aggregate_bwd(ref.view(-1, 1), h["prep"], want_bias=False)
        assert torch.equal(dh[q][:h["nb"]], rdh.view(-1)[:h["nb"]])
        ref_tot += float(ref[:h["nb"]].double().sum())
    assert abs(float(tot) - ref_tot) <= 1e-5 * max(1.0, abs(ref_tot))
    assert int(ops._ticket(torch.device("cuda", 0)).ne(0).sum()) == 0


def test_backward_aggregations_carried_by_the_sampler_heads_backward_launches():
    """ops.carry_backward_aggregations: two few-row backward aggregations of the classifier (47 columns: scalar form, 256: vector
    form, both with ReLU gate and bias gradient) recorded and carried as extra grid columns of the two launches of the sampler
    heads' backward pass (grapes_sampler_head_bwd_multi_phase 1 / 2) — every output of the four launches equal BIT FOR BIT to the
    launches on their own; a host nobody used is issued by flush_backward_hosts; tickets are left zero; a GEMM issued between the
    two phases sees the first aggregation's result."""
    _cuda()
    from grapes_amd import ops
    from grapes_amd.graph import DeviceGraph
    rng = np.random.default_rng(15)
    N = 60000
    ei = rng.integers(0, N, (2, N * 8))
    indptr, indices = O.build_csr(np.concatenate([ei, ei[::-1]], axis=1), N)
    dg = DeviceGraph.from_csr(indptr, indices)
    st = dg.status
    hops = []
    for m in (200, 450, 90):
        prev = _t(rng.permutation(N)[:m], torch.int32)
        src, dst, d_e, eoff = ops.frontier_expand_fused(dg.rowptr, dg.col, prev, 1 << 16, status=st)
        ops.bitmap_mark_hop(dg.prev_bits, dg.bits, None, prev, eoff, dst, N, d_e=d_e, status=st)
        n_cap = 40000
        batch, neigh, nbl, counts, cand_pos = ops.frontier_compact(dg.bits, None, dg.prev_bits, N, n_cap, node_map=dg.node_map,
                                                                   status=st, want_cand_pos=True)
        ops.bitmap_clear(dg.prev_bits, prev)
        prep = ops.PreparedGraph(src, dst, n_cap, d_n=counts[0:1], d_e=d_e, status=st, src_grouped=True, items_fwd=False,
                                 node_map=dg.node_map)
        hops.append(dict(logit=_t(rng.standard_normal(n_cap).astype(np.float32)), mask=_t((rng.random(n_cap) < 0.2).astype(np.float32)),
                         cand_pos=cand_pos, prep=prep))
    # the classifier's sampled subgraph: 1022 live rows of 1024, a few thousand edges among them
    n_c, n_live = 1024, 1022
    e = rng.integers(0, n_live, (2, 6000)).astype(np.int32)
    d_nc = torch.tensor([n_live], dtype=torch.int32, device="cuda")
    cprep = ops.PreparedGraph(_t(e[0], torch.int32), _t(e[1], torch.int32), n_c, d_n=d_nc, status=st)
    assert int(st) == 0
    scale = torch.tensor([0.9], device="cuda")
    d47 = _t(rng.standard_normal((n_c, 47)).astype(np.float32))
    d256 = _t(rng.standard_normal((n_c, 256)).astype(np.float32)); act256 = _t(rng.standard_normal((n_c, 256)).astype(np.float32))
    w = _t(rng.standard_normal((47, 256)).astype(np.float32))

    def head(hosted):
        tot = torch.zeros(1, device="cuda")
        hb = ops.SamplerHeadBwdMulti([h["logit"] for h in hops], [h["mask"] for h in hops], [h["cand_pos"] for h in hops],
                                     [h["prep"] for h in hops], d_grad_scale=scale, sum_out=tot)
        if not hosted:
            hb.launch(0)
            a47, b47 = ops.gcn_aggregate_bwd(d47, cprep)
            dx = ops.linear_bwd_input(a47, w, d_n=d_nc)
            a256, b256 = ops.gcn_aggregate_bwd(d256, cprep, relu_out=act256)
        else:
            ops.carry_backward_aggregations([lambda: hb.launch(1), lambda: hb.launch(2)])
            a47, b47 = ops.gcn_aggregate_bwd(d47, cprep)                      # rides in the d-logits launch
            dx = ops.linear_bwd_input(a47, w, d_n=d_nc)                        # between the phases: reads the first rider's output
            a256, b256 = ops.gcn_aggregate_bwd(d256, cprep, relu_out=act256)   # rides in the aggregation launch
            assert not ops._BWD_HOSTS
            ops.flush_backward_hosts()
        torch.cuda.synchronize()
        return [hb.dlog.clone(), hb.dh.clone(), tot.clone(), a47[:n_live].clone(), b47.clone(), dx[:n_live].clone(),
                a256[:n_live].clone(), b256.clone()]

    ref = head(False)
    got = head(True)
    for q, h in enumerate(hops):
        nb = int(h["prep"].d_n)
        assert torch.equal(got[0][q][:nb], ref[0][q][:nb]) and torch.equal(got[1][q][:nb], ref[1][q][:nb])
    for a, b in zip(got[2:], ref[2:]):
        assert torch.equal(a, b)
    assert float(ref[6].abs().sum()) > 0 and float(ref[3].abs().sum()) > 0
    # one aggregation only: the second host is issued by the flush
    tot = torch.zeros(1, device="cuda")
    hb = ops.SamplerHeadBwdMulti([h["logit"] for h in hops], [h["mask"] for h in hops], [h["cand_pos"] for h in hops],
                                 [h["prep"] for h in hops], d_grad_scale=scale, sum_out=tot)
    ops.carry_backward_aggregations([lambda: hb.launch(1), lambda: hb.launch(2)])
    a47, _ = ops.gcn_aggregate_bwd(d47, cprep)
    assert len(ops._BWD_HOSTS) == 1
    ops.flush_backward_hosts()
    torch.cuda.synchronize()
    assert torch.equal(a47[:n_live], ref[3]) and torch.equal(tot, ref[2])
    for q, h in enumerate(hops):
        nb = int(h["prep"].d_n)
        assert torch.equal(hb.dh[q][:nb], ref[1][q][:nb])
    assert int(ops._ticket(torch.device("cuda", 0)).ne(0).sum()) == 0
    assert int(st) == 0


def test_gcn_prepare_with_scratch_cleared_by_the_compaction():
    """GRAPES_PREP_PREZEROED: the graph build without its own clearing launch (the compaction before it zeroes the
    counters and csr_dst, the TensorMap relabel happens inside the per-edge kernels) == the ordinary build."""
    _cuda()
    from grapes_amd import ops
    from grapes_amd.graph import DeviceGraph
    rng = np.random.default_rng(9)
    N = 80000
    ei = rng.integers(0, N, (2, N * 10))
    indptr, indices = O.build_csr(np.concatenate([ei, ei[::-1]], axis=1), N)
    dg = DeviceGraph.from_csr(indptr, indices)
    st = dg.status
    prev = _t(rng.permutation(N)[:500], torch.int32)
    e_cap, n_cap = 1 << 15, 20000
    src, dst, d_e, eoff = ops.frontier_expand_fused(dg.rowptr, dg.col, prev, e_cap, status=st, mark_prev_bits=dg.prev_bits,
                                                    mark_bits=dg.bits, num_nodes=N)
    scr = ops.PreparedGraph.scratch(n_cap, e_cap, prev.device)
    scr[0].fill_(0x5A); scr[1].fill_(-7)                                     # dirty scratch: the compaction must clear it
    batch, neigh, nbl, counts = ops.frontier_compact(dg.bits, None, dg.prev_bits, N, n_cap, node_map=dg.node_map, status=st,
                                                     zero=scr[2])
    ops.bitmap_clear(dg.prev_bits, prev)
    a = ops.PreparedGraph(src, dst, n_cap, d_n=counts[0:1], d_e=d_e, status=st, src_grouped=True, items_fwd=False,
                          node_map=dg.node_map, head_ids=batch, scratch=scr)
    b = ops.PreparedGraph(src, dst, n_cap, d_n=counts[0:1], d_e=d_e, status=st, src_grouped=True, items_fwd=False,
                          node_map=dg.node_map, head_ids=batch)
    n, ne = int(counts[0]), int(b.rowptr_t[int(counts[0])])
    assert int(st) == 0 and n > 2048
    assert torch.equal(a.rowptr_t[:n + 1], b.rowptr_t[:n + 1]) and torch.equal(a.rowptr_s[:n + 1], b.rowptr_s[:n + 1])
    assert torch.equal(a.csr_src[:ne], b.csr_src[:ne]) and torch.equal(a.csr_dst[:ne], b.csr_dst[:ne])
    assert torch.equal(a.dinv[:n], b.dinv[:n]) and torch.equal(a.row_head[:n], b.row_head[:n])
    assert a.n_long.tolist()[:3] == b.n_long.tolist()[:3]
    assert int(ops.sync_scratch("cuda").ne(0).sum()) == 0          # look-back words and the grid barrier's pair are zero again


@pytest.mark.parametrize("rows,deg", [(500, 10), (15000, 24), (3000, 90)])
def test_gcn_prepare_one_cooperative_launch_equals_the_four_launch_build(rows, deg):
    _run_child_with_env(dict(GRAPES_PREP_FUSED="1"), "_prep_fused_case", rows, deg)


def _run_child_with_env(env, fn, *args):
    """the library reads its A/B switches once per process: run the case in a child process with the switch set"""
    import subprocess, sys
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r}); "
            f"import test_hip_parity as T; T.{fn}(*{args!r}); print('child ok')")
    e = dict(os.environ); e.update(env)
    e["GRAPES_DIAG"] = "1"          # the A/B switches exist in the diagnostic build only (grapes_amd/_lib.py)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=e)
    assert p.returncode == 0 and "child ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def _prep_fused_case(rows, deg):
    """The grouped, pre-zeroed graph build as ONE cooperative launch (prep_fused_k: counts / scan / fill / row order separated
    by grid barriers, agent-scope hand-offs) == the general four-launch build, bit for bit, over repeated launches that share
    the barrier scratch; with 15000 x 24 x 2 edges a thread owns more edges than it keeps in registers (the spill path), with
    degree 90 rows are longer than the short-row sort and the head records."""
    _cuda()
    from grapes_amd import ops
    from grapes_amd.graph import DeviceGraph
    rng = np.random.default_rng(rows)
    N = 120000
    ei = rng.integers(0, N, (2, N * deg // 2))
    ei[1, ::97] = ei[0, ::97]                                                 # some self-loops in the adjacency
    indptr, indices = O.build_csr(np.concatenate([ei, ei[::-1]], axis=1), N)
    dg = DeviceGraph.from_csr(indptr, indices)
    st = dg.status
    for rep in range(3):
        prev = _t(rng.permutation(N)[:rows], torch.int32)
        e_cap, n_cap = 1 << 20, N + 1
        eoff, d_e = ops.frontier_offsets(dg.rowptr, prev)
        src, dst, _ = ops.frontier_expand(dg.rowptr, dg.col, prev, eoff, e_cap, status=st)
        ops.bitmap_mark(dg.prev_bits, None, prev, N, status=st)
        ops.bitmap_mark_rows(dg.bits, dg.bits1, prev, eoff, N, status=st)
        ops.bitmap_mark(dg.bits, dg.bits1, dst, N, d_n=d_e, status=st)
        scr = ops.PreparedGraph.scratch(n_cap, e_cap, prev.device)
        scr[0].fill_(0x5A); scr[1].fill_(-7)
        batch, neigh, nbl, counts = ops.frontier_compact(dg.bits, dg.bits1, dg.prev_bits, N, n_cap, node_map=dg.node_map, status=st,
                                                         zero=scr[2])
        ops.bitmap_clear(dg.prev_bits, prev)
        a = ops.PreparedGraph(src, dst, n_cap, d_n=counts[0:1], d_e=d_e, status=st, src_grouped=True, items_fwd=False,
                              node_map=dg.node_map, head_ids=batch, scratch=scr)
        b = ops.PreparedGraph(src, dst, n_cap, d_n=counts[0:1], d_e=d_e, status=st, src_grouped=True, items_fwd=False,
                              node_map=dg.node_map, head_ids=batch)
        n, ne = int(counts[0]), int(b.rowptr_t[int(counts[0])])
        assert int(st) == 0 and n > 2048 and ne > 0
        assert torch.equal(a.rowptr_t[:n + 1], b.rowptr_t[:n + 1]) and torch.equal(a.rowptr_s[:n + 1], b.rowptr_s[:n + 1])
        assert torch.equal(a.csr_src[:ne], b.csr_src[:ne]) and torch.equal(a.csr_dst[:ne], b.csr_dst[:ne])
        assert torch.equal(a.dinv[:n], b.dinv[:n]) and torch.equal(a.row_head[:n], b.row_head[:n])
        assert a.n_long.tolist()[:3] == b.n_long.tolist()[:3]
        nlt, nls = a.n_long.tolist()[:2]
        key = lambda it, k: sorted(map(tuple, it[:2 * k].view(-1, 2).tolist()))
        assert key(a.items_t, nlt) == key(b.items_t, nlt) and key(a.items_s, nls) == key(b.items_s, nls)
        assert int(ops.sync_scratch("cuda").ne(0).sum()) == 0


def test_self_feeding_captured_step_equals_host_fed_steps():
    """GraphedTrainer.attach_loader + step_next (batch taken on the device by the step's first kernel, edge totals kept on
    the device) == step(batch) fed from the host with the same sequential chunks: same sampled sets and logits every step,
    and edge_totals == the sum of the per-graph counters of all steps but the last."""
    _cuda()
    from grapes_amd import synth
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.step_graph import GraphedTrainer
    n, deg, F, C, B, K, hops, H = 30000, 12.0, 100, 9, 128, 96, 2, 256
    indptr, indices = synth.synth_csr_numpy(n, deg, 2000, seed=3)
    rng = np.random.default_rng(4)
    X = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.integers(0, C, n)).cuda()
    train = torch.from_numpy(rng.permutation(n)[:700].astype(np.int64)).cuda()
    stride, offset = 2, 1

    def build():
        torch.manual_seed(0)
        c, gf, z = GCN(F, [H, C]).cuda(), GCN(F + hops + 1, [H, 1]).cuda(), GCN(F, [H, 1]).cuda()
        oc = torch.optim.Adam(c.parameters(), lr=1e-3, capturable=True)
        og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=1e-4, capturable=True)
        return GraphedTrainer(DeviceGraph.from_csr(indptr, indices), X, y, c, gf, z, batch_size=B, sampling_hops=hops,
                              num_samples=K, loss_coef=50.0, optimizer_c=oc, optimizer_gf=og, e_cap=1 << 15, philox_seed=5,
                              capture=True)
    a, b = build(), build()
    b.attach_loader(train, stride=stride, offset=offset)
    with pytest.raises(RuntimeError):
        b.step(train[:B])
    tot = torch.zeros(2 * hops, dtype=torch.int64, device="cuda")
    steps = 7
    for s in range(steps):
        o = ((s * stride + offset) * B) % max(1, train.numel() - B)
        oa = a.step(train[o:o + B])
        ob = b.step_next()
        torch.cuda.synchronize()
        b.check()
        assert torch.equal(b.targets, train[o:o + B].to(torch.int32))
        for hop in range(hops):
            kc = int(oa["kept_counts"][hop])
            assert kc == int(ob["kept_counts"][hop]) and torch.equal(oa["kept"][hop][:kc], ob["kept"][hop][:kc])
        na = int(oa["n_all"])
        assert na == int(ob["n_all"]) and torch.equal(oa["logits"][:na], ob["logits"][:na])
        assert torch.equal(oa["agg_counts"], ob["agg_counts"])
        if s < steps - 1:
            tot += ob["agg_counts"].to(torch.int64)
    # (the pipelined trainer has already taken the NEXT step's batch: its prelude runs one step ahead)
    assert torch.equal(b.edge_totals, tot) and int(b._cursor) == steps + (1 if b._sets is not None else 0)


@pytest.mark.parametrize("sizes", [(256, 256, 256, 256), (1, 0, 3, 0), (700, 1500, 1800, 0), (256, 512, 512), (5,)])
def test_union_sorted_equals_bitmap_mark_and_compact(sizes):
    """all_nodes by one workgroup (bitonic sort + unique) == marking the lists into the N-bit map and compacting it:
    ascending ids, duplicates across and inside lists, device-side counts below capacity, TensorMap and slice-mark
    clearing side effects."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(sum(sizes))
    N = 300000
    pool = rng.permutation(N)[:1500]                                  # a small pool => many duplicates
    lists, live = [], []
    for k, sz in enumerate(sizes):
        if sz == 0:
            continue
        ids = _t(pool[rng.integers(0, len(pool), sz + 7)], torch.int32)
        lists.append((ids, torch.tensor([sz], dtype=torch.int32, device="cuda")))
        live.append(ids[:sz].cpu().numpy())
    ref = np.unique(np.concatenate(live))
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    mult = torch.ones(N, dtype=torch.int32, device="cuda"); node_map = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    out, counts = ops.union_sorted(lists, N, len(ref) + 5, node_map=node_map, status=st, unmark_mult=mult)
    c = int(counts[0])
    assert c == len(ref) and np.array_equal(out[:c].cpu().numpy(), ref) and int(st) == 0
    assert np.array_equal(node_map[torch.from_numpy(ref).cuda().long()].cpu().numpy(), np.arange(c))
    assert int((mult == 0).sum()) == c and int(mult[torch.from_numpy(ref).cuda().long()].sum()) == 0
    W = (N + 63) // 64
    bits = torch.zeros(W, dtype=torch.int64, device="cuda")
    ops.bitmap_mark_lists(bits, None, lists, N, status=st)
    b, _, _, cc = ops.frontier_compact(bits, None, None, N, len(ref) + 5, status=st)
    assert int(cc[0]) == c and torch.equal(b[:c], out[:c])
    # capacity overflow is flagged
    if c > 1:
        ops.union_sorted(lists, N, c - 1, status=st)
        assert int(st) & 2


@pytest.mark.parametrize("n,cap,H,deg", [(30000, 30000, 256, 40), (9000, 12000, 64, 8), (1500, 1500, 256, 6)])
def test_rank1_backward_aggregation_equals_the_three_launch_path(n, cap, H, deg):
    """Backward of (transform-first GCNConv -> ReLU -> 1-wide head) — Reddit's and Cora's sampler / log-Z nets — without
    the outer product and its masked copy (grapes_gcn_aggregate_bwd_rank1): dW2, db1 and (at the step's width, 256) dH equal, BIT FOR BIT, what
    linear_bwd_weight (dh2ᵀ act) + linear_bwd_input (dh2 ⊗ w2) + gcn_aggregate_bwd (mask, bias gradient, Âᵀ) produce; against
    fp64 autograd of the same expression within 1e-5.  Hub sources (rows cut into items) and a live count below the capacity."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(n + H)
    m_src = 300                                             # frontier-like: few sources, many targets, some hubs
    wts = rng.pareto(1.1, m_src) + 1
    degs = np.minimum(n, np.maximum(1, (wts / wts.sum() * n * deg / 8).astype(np.int64)))
    srcs = np.sort(rng.permutation(n)[:m_src])
    src = np.repeat(srcs, degs)
    dst = np.concatenate([np.sort(rng.permutation(n)[:d]) for d in degs])
    ls, ld = _t(src, torch.int32), _t(dst, torch.int32)
    d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    prep = ops.PreparedGraph(ls, ld, cap, d_n=d_n, src_grouped=True, items_fwd=False)
    torch.manual_seed(n)
    act = torch.relu(torch.randn(cap, H, device="cuda"))
    dh2 = torch.randn(cap, 1, device="cuda")
    w2 = torch.randn(1, H, device="cuda") * 0.3
    for accumulate in (False, True):
        init = 3.0 if accumulate else float("nan")
        # three-launch path
        dw2_a = torch.full((1, H), init, device="cuda"); db1_a = torch.full((H,), init, device="cuda")
        ops.linear_bwd_weight(dh2, act, d_n=d_n, out=dw2_a, accumulate=accumulate)
        dact = ops.linear_bwd_input(dh2, w2, d_n=d_n)
        dh_a, _ = ops.gcn_aggregate_bwd(dact, prep, relu_out=act, dbias=db1_a, accumulate_bias=accumulate)
        # one pass
        dw2_b = torch.full((1, H), init, device="cuda"); db1_b = torch.full((H,), init, device="cuda")
        dh_b = ops.gcn_aggregate_bwd_rank1(act, dh2.view(-1), w2.view(-1), prep, dw_head=dw2_b.view(-1), dbias=db1_b,
                                           accumulate=accumulate)
        if H > 128 and cap > 2048:      # (narrower rows: the three-launch path aggregates with lanes in slots; few rows: in its one-launch
            assert torch.equal(dh_a[:n], dh_b[:n])          #  small-graph kernel — other, equally fixed, summation orders)
        else:
            assert _close(dh_b[:n].cpu().numpy(), dh_a[:n].cpu().numpy(), 2e-6)
        assert torch.equal(dw2_a, dw2_b)
        if cap > 2048:
            assert torch.equal(db1_a, db1_b)
        else:            # (few rows: the three-launch path sums the bias gradient inside its one-launch small-graph kernel)
            assert _close(db1_b.cpu().numpy(), db1_a.cpu().numpy(), 2e-6)
    # fp64 autograd of  loss = sum(dh2_up * (Â (relu(pre) w2ᵀ)))  reduces to the same three quantities
    a64 = act[:n].double().cpu(); d64 = dh2[:n].double().cpu().view(-1); w64 = w2.double().cpu().view(-1)
    dpre = (a64 > 0).double() * d64[:, None] * w64[None, :]
    assert _close(dw2_b.cpu().numpy().reshape(-1) - 3.0, (d64[:, None] * a64).sum(0).numpy(), 1e-5)
    assert _close(db1_b.cpu().numpy() - 3.0, dpre.sum(0).numpy(), 1e-5)
    dinv = prep.dinv[:n].double().cpu()
    ref = dinv[:, None] ** 2 * dpre
    nl = src != dst                                         # (existing loops are replaced by the unit loop: gcn_norm)
    ts, td = torch.from_numpy(src[nl]), torch.from_numpy(dst[nl])
    ref.index_add_(0, ts, (dinv[ts] * dinv[td])[:, None] * dpre[td])
    assert _close(dh_b[:n].cpu().numpy(), ref.numpy(), 1e-5)


def test_dropout_with_p_one_drops_everything():
    """F.dropout accepts p = 1 and returns zeros: the Philox kernels must too — forward zeros, keep mask empty,
    backward zeros, no inf / NaN from the 1 / (1 - p) scale."""
    _cuda()
    from grapes_amd import ops
    x = torch.randn(300, 48, device="cuda")
    y, keep = ops.dropout_fwd(x, 1.0, philox_seed=5, philox_offset=0)
    assert float(y.abs().max()) == 0.0 and int(keep.sum()) == 0
    dx = ops.dropout_bwd(torch.randn_like(x), keep, 1.0)
    assert float(dx.abs().max()) == 0.0 and bool(torch.isfinite(dx).all())
    y0, keep0 = ops.dropout_fwd(x, 0.0, philox_seed=5, philox_offset=0)
    assert torch.equal(y0, x) and int(keep0.sum()) == x.numel()


def _compact_dense_block_case():
    """one workgroup of the 512-thread compaction sees 512 x 64 = 32768 ids that are ALL new neighbours: the packed
    (count | count << 16) scan must read its upper half unsigned"""
    from grapes_amd import ops
    from grapes_amd.graph import DeviceGraph
    N = 200_000
    dg = DeviceGraph.from_csr(np.zeros(N + 1, dtype=np.int64), np.zeros(0, dtype=np.int32))
    ids = torch.arange(0, 70_000, dtype=torch.int32, device="cuda")          # two full 32768-id blocks and a part
    ops.bitmap_mark(dg.bits, dg.bits1, ids, N, status=dg.status)
    batch, neigh, nbl, counts = ops.frontier_compact(dg.bits, dg.bits1, dg.prev_bits, N, N + 1, node_map=dg.node_map, status=dg.status)
    assert counts.tolist() == [70_000, 70_000] and int(dg.status) == 0
    assert torch.equal(batch[:70_000], ids) and torch.equal(neigh[:70_000], ids)
    assert torch.equal(nbl[:70_000], torch.arange(70_000, dtype=torch.int32, device="cuda"))
    assert int(ops.sync_scratch("cuda").ne(0).sum()) == 0


def test_compaction_with_512_thread_workgroups_full_of_new_neighbours():
    _cuda()
    _run_child_with_env(dict(GRAPES_COMPACT_THREADS="512"), "_compact_dense_block_case")


@pytest.mark.parametrize("F", [32, 48, 64, 100, 128, 256])
def test_prescaled_full_graph_aggregation_matches_the_weighted_form(F):
    """Full-batch inference form (grapes_gcn_aggregate_fwd_prescaled over rows scaled by their own dinv) == Â h + b of the
    training kernels to fp32 rounding, for every lane layout (8 / 16 / 32 lanes per row and the 256-wide kernel), with hub rows
    that go through the item (chunk + combine) path, bias and ReLU."""
    _cuda()
    from grapes_amd import ops, synth
    from grapes_amd.graph import DeviceGraph
    n = 20000
    indptr, indices = synth.synth_csr_numpy(n, 16.0, 3000, seed=F)
    g = DeviceGraph.from_csr(indptr, indices)
    prep = g.gcn_prepared()
    assert int((prep.rowptr_t[1:] - prep.rowptr_t[:-1]).max()) > 256          # hubs: several 64-entry items
    torch.manual_seed(F)
    h = torch.randn(n, F, device="cuda")
    b = torch.randn(F, device="cuda")
    for bias, relu in ((None, False), (b, True)):
        ref = ops.gcn_aggregate_fwd(h, prep, bias, relu)
        out = ops.gcn_aggregate_fwd_prescaled(ops.scale_rows(h, prep.dinv), prep, bias, relu)
        assert _close(out.cpu().numpy(), ref.cpu().numpy(), 2e-6)
    hs = h.clone()
    ops.scale_rows(hs, prep.dinv, out=hs)                                       # in place
    assert torch.equal(hs, ops.scale_rows(h, prep.dinv))


@pytest.mark.parametrize("n,fi,fo", [(20000, 256, 256), (7001, 256, 48), (513, 100, 256), (130, 37, 6)])
def test_row_scaled_linear_equals_linear_then_scale_rows(n, fi, fo):
    """grapes_linear_fwd_row_scaled (dinv scaling in the GEMM epilogue, the full-batch inference transform: eval.py:50 via
    modules/gcn.py:32) returns the bits of grapes_linear_fwd followed by grapes_scale_rows — aligned and unaligned widths, a
    ragged last tile, a device-side row count."""
    _cuda()
    from grapes_amd import ops
    torch.manual_seed(n + fo)
    x = torch.randn(n, fi, device="cuda")
    w = torch.randn(fo, fi, device="cuda") / fi ** 0.5
    sc = torch.rand(n, device="cuda") + 0.01
    h = ops.linear_fwd(x, w)
    ref = (h * sc[:, None]) if fo % 4 else ops.scale_rows(h, sc)
    out = ops.linear_fwd_row_scaled(x, w, sc)
    assert torch.equal(out, ref)
    d_n = torch.tensor([n - 37], dtype=torch.int32, device="cuda")
    out2 = torch.full((n, fo), 7.0, device="cuda")
    ops.linear_fwd_row_scaled(x, w, sc, d_n=d_n, out=out2)
    assert torch.equal(out2[: n - 37], ref[: n - 37])


def test_staged_slice_equals_slice_filter_with_duplicate_columns():
    """slice_adjacency through the expansion's stage + the classifier graph build's assembly (no slice launch) == the
    slice_filter kernels: same edge list in the same order, multiplicities > 1 (duplicate column ids, as in golden G2)
    included, ragged last wavefront-block, and the graphs built from the two lists are identical."""
    _cuda()
    from grapes_amd import ops, synth
    from grapes_amd.graph import DeviceGraph
    rng = np.random.default_rng(11)
    N = 50000
    indptr, indices = synth.synth_csr_numpy(N, 20.0, 4000, seed=5)
    dg = DeviceGraph.from_csr(indptr, indices)
    st = dg.status
    rows = _t(rng.permutation(N)[:700], torch.int32)
    cols_np = rng.permutation(N)[:900]
    cols_np = np.concatenate([cols_np, cols_np[:60], cols_np[:20]])              # some columns twice, some three times
    cols = _t(cols_np, torch.int32)
    ops.slice_mark(dg.mult, cols)
    e_cap = 1 << 16
    stage = ops.slice_stage(e_cap, "cuda"); stage.fill_(-123456)                 # dirty: nothing may depend on its contents
    src, dst, d_e, eoff = ops.frontier_expand_fused(dg.rowptr, dg.col, rows, e_cap, status=st, count_mult=dg.mult, slice_stage=stage)
    e = int(d_e)
    assert 0 < e < e_cap and e % 64 != 0 and int(st) == 0
    kcap = 1 << 14
    rs, rd, rc = ops.slice_filter(dg.mult, src, dst, kcap, d_e=d_e, status=st)
    m = int(rc)
    assert m > 100 and bool((dg.mult[rd[:m].long()] > 1).any())
    # node set for the small graphs: the ids that occur in the slice, ascending
    ids = torch.unique(torch.cat([rs[:m], rd[:m]])).to(torch.int32)
    nloc = ids.numel()
    assert nloc <= 2048
    dg.node_map[ids.long()] = torch.arange(nloc, dtype=torch.int32, device="cuda")
    d_n = torch.tensor([nloc], dtype=torch.int32, device="cuda")
    ks = torch.full((kcap,), -1, dtype=torch.int32, device="cuda"); kd = torch.full((kcap,), -1, dtype=torch.int32, device="cuda")
    kc = torch.zeros(1, dtype=torch.int32, device="cuda")
    a = ops.PreparedGraph.small_batch([(ks, kd, kc)], 2048, d_n=d_n, status=st, node_map=dg.node_map, stages=[(stage, d_e, e_cap)])[0]
    assert int(kc) == m and torch.equal(ks[:m], rs[:m]) and torch.equal(kd[:m], rd[:m])
    b = ops.PreparedGraph.small_batch([(rs, rd, rc)], 2048, d_n=d_n, status=st, node_map=dg.node_map)[0]
    ne = int(b.rowptr_t[nloc])
    assert int(st) == 0 and torch.equal(a.rowptr_t[:nloc + 1], b.rowptr_t[:nloc + 1]) and torch.equal(a.csr_src[:ne], b.csr_src[:ne])
    assert torch.equal(a.rowptr_s[:nloc + 1], b.rowptr_s[:nloc + 1]) and torch.equal(a.csr_dst[:ne], b.csr_dst[:ne]) and torch.equal(a.dinv[:nloc], b.dinv[:nloc])
    ops.slice_mark(dg.mult, cols, unmark=True)


@pytest.mark.parametrize("m", [300, 1500])
def test_counted_hop_build_equals_the_four_launch_build(m):
    """The hop graph built with its degree counting folded into the expansion (per-edge in-degree atomics that return the entry's
    slot) and the compaction (row starts, dinv, segments, long-row items) + two launches, against grapes_gcn_prepare's four
    launches over the same expansion: every array the consumers read is identical — CSRs by target and by source, dinv, head
    records, the edge count, the long-row work items (as sets) — with a hub in the query list (out-degree in the thousands,
    rows of hundreds of entries), existing self-loops (replaced by the unit loop) and query nodes that neighbour each other;
    the counter tables are zero again afterwards."""
    _cuda()
    from grapes_amd import ops
    from grapes_amd.graph import DeviceGraph
    rng = np.random.default_rng(77 + m)
    n = 60000
    hub = np.stack([rng.permutation(n)[:5000], np.full(5000, 11, np.int64)])
    rnd = rng.integers(0, n, (2, 250000))
    loops = np.stack([np.arange(0, n, 7), np.arange(0, n, 7)])
    indptr, indices = O.build_csr(np.concatenate([hub, hub[::-1], rnd, rnd[::-1], loops], axis=1), n)
    g = DeviceGraph.from_csr(indptr, indices)
    prev = rng.permutation(n)[:m].astype(np.int32)
    prev[0], prev[1] = 11, 7
    prev = np.unique(prev).astype(np.int32)                      # a query list holds each node once (order is free)
    rng.shuffle(prev)
    prev_t = _t(prev, torch.int32)
    e_cap = 1 << 18
    n_cap = e_cap + len(prev) + 1
    hc = g.hop_counters()

    def run(counted, cursor_form=False):
        hb = ops.HopBuild(n_cap, e_cap, "cuda", cursor_form=cursor_form) if counted else None
        src, dst, d_e, eoff = ops.frontier_expand_fused(g.rowptr, g.col, prev_t, e_cap, status=g.status, mark_prev_bits=g.prev_bits,
                                                        mark_bits=g.bits, num_nodes=n, count=(hc, hb) if counted else None)
        pscr = None if counted else ops.PreparedGraph.scratch(n_cap, e_cap, "cuda")
        batch, neigh, nbl, counts = ops.frontier_compact(g.bits, None, g.prev_bits, n, n_cap, node_map=g.node_map, status=g.status,
                                                         zero=[(hb.csr_dst, e_cap)] if counted else list(pscr[2]),
                                                         degrees=(hc, hb) if counted else None)
        if counted:
            prep = ops.PreparedGraph.counted(src, dst, hb, n_cap, counts[0:1], d_e, g.node_map, status=g.status, head_ids=batch)
        else:
            prep = ops.PreparedGraph(src, dst, n_cap, d_n=counts[0:1], d_e=d_e, status=g.status, src_grouped=True, items_fwd=False,
                                     node_map=g.node_map, head_ids=batch, scratch=pscr)
        torch.cuda.synchronize()
        g.prev_bits.zero_()
        assert int(g.status.item()) == 0
        nb = int(counts[0])
        E = int(prep.rowptr_t[nb])
        nl = prep.n_long.cpu().numpy()
        items = lambda half, k: sorted(map(tuple, half[: 2 * k].view(-1, 2).cpu().numpy().tolist()))
        return dict(nb=nb, E=E, batch=batch[:nb].clone(), rt=prep.rowptr_t[: nb + 1].clone(), rs=prep.rowptr_s[: nb + 1].clone(),
                    dinv=prep.dinv[:nb].clone(), cs=prep.csr_src[:E].clone(), cd=prep.csr_dst[:E].clone(),
                    head=prep.row_head[:nb].clone(), n_edges=int(nl[2]), it=items(prep.items_t, int(nl[0])),
                    is_=items(prep.items_s, int(nl[1])), e_list=int(d_e))

    a, b = run(False), run(True)
    assert a["nb"] == b["nb"] and a["E"] == b["E"] and a["n_edges"] == b["n_edges"] == a["E"] and a["E"] < a["e_list"]   # (loops dropped)
    for k in ("batch", "rt", "rs", "dinv", "cs", "cd", "head"):
        assert torch.equal(a[k], b[k]), k
    assert a["it"] == b["it"] and a["is_"] == b["is_"] and len(a["is_"]) > 0 and (m < 1000 or len(a["it"]) > 0)
    lens = (a["rt"][1:] - a["rt"][:-1]).cpu().numpy()
    assert (lens == 0).any() and (m < 1000 or lens.max() > 64)
    for t in (hc.indeg, hc.loops, hc.wsum, hc.sync2, ops.sync_scratch("cuda")):
        assert int(t.abs().max()) == 0
    # an expansion that overflows its edge capacity (flagged) leaves the counter tables clean for the next one
    hb_small = ops.HopBuild(4096 + len(prev) + 1, 4096, "cuda")
    ops.frontier_expand_fused(g.rowptr, g.col, prev_t, 4096, status=g.status, mark_prev_bits=g.prev_bits, mark_bits=g.bits,
                              num_nodes=n, count=(hc, hb_small))
    ops.frontier_compact(g.bits, None, g.prev_bits, n, 4096 + len(prev) + 1, node_map=g.node_map, status=g.status,
                         degrees=(hc, hb_small))
    torch.cuda.synchronize()
    assert int(g.status.item()) & 1                              # GRAPES_STATUS_EDGE_OVERFLOW
    g.status.zero_(); g.prev_bits.zero_()
    for t in (hc.indeg, hc.loops, hc.wsum, hc.sync2, g.bits):
        assert int(t.abs().max()) == 0
    c = run(True, cursor_form=True)                              # the cursor form (the fill's own atomics), on the zero-again tables
    for k in ("batch", "rt", "rs", "dinv", "cs", "cd", "head"):
        assert torch.equal(a[k], c[k]), k
    assert a["is_"] == c["is_"] and a["it"] == c["it"]
    for t in (hc.indeg, hc.loops, hc.wsum, hc.sync2):
        assert int(t.abs().max()) == 0


def test_two_first_layers_in_one_launch_equal_two_launches():
    """grapes_linear_relu_head_fwd_bits_pair (the sampler net's and the log-Z net's first layers over the same hop-0 rows, the second
    reading the leading columns of the first's input) against two grapes_linear_relu_head_fwd_bits calls: gate words and head
    outputs bit for bit; row counts with a ragged last panel, fewer panels than workgroups, and a device-side count."""
    _cuda()
    from grapes_amd import ops
    torch.manual_seed(41)
    for n, cap, dn, Ka, Kb in ((12611, 12611, None, 104, 100), (4099, 9000, 4099, 104, 100), (37000, 37000, None, 104, 100),
                               (3636, 3636, None, 132, 128), (5001, 6000, 5001, 132, 100)):      # (arxiv: 9 and 8 K steps in one instance)
        H = 256
        x = torch.randn(cap, Ka, device="cuda")
        xb = x[:, :Kb]
        wa = (torch.randn(H, Ka, device="cuda") * 0.2).contiguous(); wb = (torch.randn(H, Kb, device="cuda") * 0.2).contiguous()
        ba, bb = torch.randn(H, device="cuda") * 0.1, torch.randn(H, device="cuda") * 0.1
        ha, hb = torch.randn(1, H, device="cuda") * 0.3, torch.randn(1, H, device="cuda") * 0.3
        d_n = None if dn is None else torch.tensor([dn], dtype=torch.int32, device="cuda")
        r = ops.linear_relu_head_fwd_bits_pair(x, wa, ba, ha, xb, wb, bb, hb, d_n=d_n)
        assert r is not None
        a1, h1 = ops.linear_relu_head_fwd_bits(x, wa, ba, ha, d_n=d_n)
        a2, h2 = ops.linear_relu_head_fwd_bits(xb, wb, bb, hb, d_n=d_n)
        m = n if dn is None else dn
        assert torch.equal(r[0].words[:m], a1.words[:m]) and torch.equal(r[1][:m], h1[:m])
        assert torch.equal(r[2].words[:m], a2.words[:m]) and torch.equal(r[3][:m], h2[:m])


def test_gate_bit_weight_gradient_in_the_parameters_own_layout():
    """grapes_linear_bwd_weight_bits_multi_cols / _pair_cols: the slab sum writes dW as [f_out, K] when the layer's operands carry
    the 4-padded width (ogbn-arxiv: 128 + 3 indicators = 131 -> 132, a zero pad column in x and w1) — bit for bit the leading K
    columns of the padded gradient, overwriting and accumulating; the pad column's slot does not exist."""
    _cuda()
    from grapes_amd import ops
    torch.manual_seed(43)
    for K, Kb in ((131, 128), (101, 100), (130, 132 - 4)):
        Kp, H, n = (K + 3) // 4 * 4, 256, 9000
        x = torch.randn(n, Kp, device="cuda"); x[:, K:] = 0
        w = (torch.randn(H, Kp, device="cuda") * 0.2).contiguous(); w[:, K:] = 0
        b = torch.randn(H, device="cuda") * 0.1; w2 = torch.randn(1, H, device="cuda") * 0.3
        d_n = torch.tensor([n - 11], dtype=torch.int32, device="cuda")
        bits, _ = ops.linear_relu_head_fwd_bits(x, w, b, w2, d_n=d_n)
        rs = torch.randn(n, device="cuda")
        for acc in (False, True):
            pad = torch.full((H, Kp), 2.0, device="cuda"); own = torch.full((H, K), 2.0, device="cuda")
            dbp, dbo = torch.full((H,), 2.0, device="cuda"), torch.full((H,), 2.0, device="cuda")
            dhp, dho = torch.full((H,), 2.0, device="cuda"), torch.full((H,), 2.0, device="cuda")
            ops.linear_bwd_weight_bits_multi([bits], [x], [rs], [d_n], w2.view(-1), w, b, pad, dbias=dbp, dw_head=dhp, accumulate=acc)
            ops.linear_bwd_weight_bits_multi([bits], [x], [rs], [d_n], w2.view(-1), w, b, own, dbias=dbo, dw_head=dho, accumulate=acc)
            assert torch.equal(own, pad[:, :K]) and torch.equal(dbo, dbp) and torch.equal(dho, dhp)
        # ... and as layer a of the pair launch (layer b: the leading Kb columns, its own dense gradient)
        xb = x[:, :Kb]; wb = (torch.randn(H, Kb, device="cuda") * 0.2).contiguous(); bb = torch.randn(H, device="cuda") * 0.1
        w2b = torch.randn(1, H, device="cuda") * 0.3
        bits_b, _ = ops.linear_relu_head_fwd_bits(xb, wb, bb, w2b, d_n=d_n)
        rsb = torch.randn(n, device="cuda")
        outs = []
        for cols in (Kp, K):
            t = [torch.full((H, cols), 3.0, device="cuda"), torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda"),
                 torch.zeros(H, Kb, device="cuda"), torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")]
            ops.linear_bwd_weight_bits_pair([bits], [x], [rs], [d_n], w2.view(-1), w, b, t[0], t[1], t[2],
                                            bits_b, xb, rsb, d_n, w2b.view(-1), wb, bb, t[3], t[4], t[5])
            outs.append(t)
        assert torch.equal(outs[1][0], outs[0][0][:, :K])
        for q in range(1, 6):
            assert torch.equal(outs[1][q], outs[0][q])
    with pytest.raises(ValueError):
        ops.linear_bwd_weight_bits_multi([bits], [x], [rs], [d_n], w2.view(-1), w, b, torch.zeros(H, Kp - 4, device="cuda"))


def test_gate_bit_weight_gradient_refuses_a_row_set_of_four_gibibytes():
    """gemm_dw_split_k<true> addresses a row set with 32-bit byte offsets from its base (include/grapes_hip.h): a row set whose
    capacity x row stride reaches 4 GiB is refused (GRAPES_EINVAL), not wrapped — nothing is read beyond the 64 live rows here,
    so the large operands are never initialised."""
    _cuda()
    from grapes_amd import ops, _lib
    H, K, ld, n_cap = 256, 104, 128, 1 << 23                                   # 2^23 rows x 128 floats x 4 B = 4 GiB
    big = torch.empty((n_cap, ld), device="cuda")
    x = big[:, :K]
    w = (torch.randn(H, K, device="cuda") * 0.2).contiguous(); b = torch.zeros(H, device="cuda"); w2 = torch.randn(H, device="cuda")
    bits = ops.GateBits(torch.zeros((n_cap, H // 32), dtype=torch.int32, device="cuda"), n_cap, H)
    rs = torch.zeros(n_cap, device="cuda")
    d_n = torch.tensor([64], dtype=torch.int32, device="cuda")
    dw = torch.zeros(H, K, device="cuda")
    with pytest.raises(_lib.GrapesHipError):
        ops.linear_bwd_weight_bits_multi([bits], [x], [rs], [d_n], w2, w, b, dw, dbias=torch.zeros(H, device="cuda"),
                                         dw_head=torch.zeros(H, device="cuda"))
    # one row less than 4 GiB is taken
    x2 = big[: n_cap - 1, :K]
    bits2 = ops.GateBits(bits.words[: n_cap - 1], n_cap - 1, H)
    big[:64].zero_()
    ops.linear_bwd_weight_bits_multi([bits2], [x2], [rs[: n_cap - 1]], [d_n], w2, w, b, dw, dbias=torch.zeros(H, device="cuda"),
                                     dw_head=torch.zeros(H, device="cuda"))
    torch.cuda.synchronize()
    assert float(dw.abs().max()) == 0.0


def _wide_compaction_case(counted):
    """frontier_compact over a 1.2M-node bitmap with sparse and dense stretches, previous-node bits, indicator marks, scratch
    clears and (counted) the degree outputs; prints nothing, leaves a digest of every output in a file named by the environment."""
    _cuda()
    import hashlib
    from grapes_amd import ops
    from grapes_amd.graph import DeviceGraph
    rng = np.random.default_rng(5)
    N = 1_200_000
    ei = rng.integers(0, N, (2, 2_000_000))
    ei[:, :200_000] = rng.integers(0, 6000, (2, 200_000))                      # a dense corner: words with many set bits
    ei[1, ::53] = ei[0, ::53]                                                  # self-loops
    indptr, indices = O.build_csr(np.concatenate([ei, ei[::-1]], axis=1), N)
    g = DeviceGraph.from_csr(indptr, indices)
    prev = np.unique(np.concatenate([rng.integers(0, 6000, 900), rng.integers(0, N, 900)])).astype(np.int32)
    rng.shuffle(prev)
    prev_t = _t(prev, torch.int32)
    e_cap = 1 << 18
    n_cap = e_cap + len(prev) + 1
    hc = g.hop_counters()
    hb = ops.HopBuild(n_cap, e_cap, "cuda") if counted else None
    epoch = 5
    src, dst, d_e, eoff = ops.frontier_expand_fused(g.rowptr, g.col, prev_t, e_cap, status=g.status, mark_prev_bits=g.prev_bits,
                                                    mark_bits=g.bits, num_nodes=N, count=(hc, hb) if counted else None)
    zt = torch.ones(70_000, dtype=torch.int32, device="cuda")
    batch, neigh, nbl, counts, cand = ops.frontier_compact(g.bits, None, g.prev_bits, N, n_cap, node_map=g.node_map, status=g.status,
                                                           ind_code=g.ind_code, epoch=epoch, ind_bit=2, want_cand_pos=True,
                                                           zero=[(zt, zt.numel())], degrees=(hc, hb) if counted else None)
    torch.cuda.synchronize()
    assert int(g.status.item()) == 0 and int(zt.abs().max()) == 0 and int(g.bits.ne(0).sum()) == 0
    nb, nn = int(counts[0]), int(counts[1])
    h = hashlib.sha256()
    parts = [batch[:nb], neigh[:nn], nbl[:nn], cand[:nb], g.node_map[batch[:nb].long()], g.ind_code[batch[:nb].long()], counts]
    if counted:
        parts += [hb.rowptr_t[: nb + 1], hb.rowptr_s[: nb + 1], hb.dinv[:nb].view(torch.int32), hb.n_long[2:3]]
        for t in (hc.indeg, hc.loops, hc.wsum, hc.sync2):
            assert int(t.abs().max()) == 0
    for t in parts:
        h.update(t.contiguous().cpu().numpy().tobytes())
    assert nb > 5_000 and nn > 5_000
    open(os.environ["GRAPES_TEST_DIGEST_FILE"], "w").write(h.hexdigest())


@pytest.mark.parametrize("counted", [False, True])
def test_eight_words_per_thread_compaction_equals_the_one_word_form(counted, tmp_path):
    """compact_emit_wide_k (a thread owns eight consecutive bitmap words: papers100M's one-launch form) against compact_emit_k on the
    same 1.2M-node frontier — every output and side job, with and without the degree outputs of the counted build.  The library
    picks the kernel once per process: two child processes, GRAPES_COMPACT_WIDE=2 forcing the wide kernel in one."""
    _cuda()
    digests = []
    for force in ("1", "2"):
        f = str(tmp_path / f"digest_{force}")
        _run_child_with_env({"GRAPES_COMPACT_WIDE": force, "GRAPES_TEST_DIGEST_FILE": f}, "_wide_compaction_case", counted)
        digests.append(open(f).read())
    assert digests[0] == digests[1] and len(digests[0]) == 64


def _tsplit_dw_case():
    """dW = dHᵀ feat(ids) on the bf16 pipe at a Reddit-like shape (30k gathered rows x 605 -> 256, masked indicator bits, a ragged
    last chunk); leaves a digest of the gradient in the file named by the environment."""
    _cuda()
    import hashlib
    from grapes_amd import ops
    rng = np.random.default_rng(8)
    N, F, ni, fo, n = 50_000, 602, 3, 256, 30_011
    X = _t(rng.standard_normal((N, F)).astype(np.float32))
    Xp, _ = ops.pad_features(X)
    ids = _t(rng.integers(0, N, n + 13), torch.int32)
    d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    code = _t(((7 << 8) | rng.integers(0, 8, N)).astype(np.int32))
    dh = _t(rng.standard_normal((n + 13, fo)).astype(np.float32))
    dW = torch.zeros((fo, F + ni), device="cuda")
    ops.linear_bwd_weight_gathered(dh, Xp, F, ids, dW, code, 7, ni, d_n=d_n, ind_mask=5, split=True)
    torch.cuda.synchronize()
    open(os.environ["GRAPES_TEST_DIGEST_FILE"], "w").write(hashlib.sha256(dW.cpu().numpy().tobytes()).hexdigest())


def test_dw_kernel_with_eight_consumer_wavefronts_is_bit_identical_to_four(tmp_path):
    """gemm_tsplit_dw_k<8> (768 threads: eight MFMA wavefronts that also stage the dH image) against gemm_tsplit_dw_k<4>: the same
    products in the same order per accumulator — equal gradients bit for bit (two child processes: the library reads
    GRAPES_TSPLIT_DW_CW once)."""
    _cuda()
    digests = []
    for cw in ("4", "8"):
        f = str(tmp_path / f"dw_{cw}")
        _run_child_with_env({"GRAPES_TSPLIT_DW_CW": cw, "GRAPES_TEST_DIGEST_FILE": f}, "_tsplit_dw_case")
        digests.append(open(f).read())
    assert digests[0] == digests[1] and len(digests[0]) == 64


def test_dw_kernel_with_swapped_operand_roles_is_bit_identical(tmp_path):
    """gemm_tsplit_dw_sw_k (the 128-wide LDS image holds feature columns, the 256-wide one all of dH: 5 tiles per slab for Reddit's
    608 columns instead of 6) against gemm_tsplit_dw_k<8> at that shape — same slabs, same K order, the same six products in the
    same order per accumulator: with the same slab count equal gradients bit for bit (GRAPES_TSPLIT_DW_SWAP=0 / 1 in the diagnostic build; the product
    library takes the swapped form whenever it needs fewer tiles)."""
    _cuda()
    digests = []
    # (the slab count follows the tile count — 768 workgroups' worth — and sets the order of the final sums: both runs get 30 slabs)
    for sw, wgs in (("0", "180"), ("1", "150")):
        f = str(tmp_path / f"dw_sw_{sw}")
        _run_child_with_env({"GRAPES_TSPLIT_DW_SWAP": sw, "GRAPES_TSPLIT_DW_WGS": wgs, "GRAPES_TEST_DIGEST_FILE": f}, "_tsplit_dw_case")
        digests.append(open(f).read())
    assert digests[0] == digests[1] and len(digests[0]) == 64


def test_weight_gradients_of_several_problems_in_one_launch():
    """grapes_linear_bwd_weight_gathered_split_multi (the sampler net's first-layer dW at two hops — one gradient, different rows and
    indicator masks — and the log-Z net's, in ONE launch whose slab budget follows the live row counts + one slab sum) against the
    three one-problem calls and against fp64: each output within 1e-6 of its own sum |dh||x| of fp64 and no further from fp64 than
    the one-problem launches (+10 %); the parameter-layout gradient ([f_out, 605]: no padding column to write) and the padded one;
    a problem without live rows contributes nothing; live counts that differ from the capacities."""
    _cuda()
    from grapes_amd import ops
    rng = np.random.default_rng(21)
    N, F, ni, fo = 40_000, 602, 3, 256
    X_np = rng.standard_normal((N, F)).astype(np.float32)
    X = _t(X_np)
    Xp, _ = ops.pad_features(X)
    epoch = 9
    code_np = ((epoch << 8) | rng.integers(0, 8, N)).astype(np.int32)
    code = _t(code_np)
    caps, lives = (9000, 30000, 9000), (7013, 21877, 7013)
    probs_np = []
    for cap, n in zip(caps, lives):
        probs_np.append((rng.integers(0, N, cap), rng.standard_normal((cap, fo)).astype(np.float32), n))
    assert ops.split_gathered_available(fo)

    def problems(dw_gf, dw_z, lives_now):
        out = []
        for q, ((ids, dh, _), n) in enumerate(zip(probs_np, lives_now)):
            z = q == 2
            out.append(dict(dh=_t(dh), ids=_t(ids, torch.int32), dw=dw_z if z else dw_gf, ind_code=None if z else code,
                            num_ind=0 if z else ni, d_n=torch.tensor([n], dtype=torch.int32, device="cuda"),
                            accumulate=(q == 1), ind_mask=0 if z else (3 if q == 0 else 7), split=True))
        return out

    for lives_now in (lives, (0, 21877, 7013)):
        for layout in ("own", "padded"):
            shp_gf = (fo, F + ni) if layout == "own" else (fo, (F + ni + 3) // 4 * 4)
            shp_z = (fo, F) if layout == "own" else (fo, (F + 3) // 4 * 4)
            a_gf, a_z = torch.full(shp_gf, 5.0, device="cuda"), torch.full(shp_z, 5.0, device="cuda")
            b_gf, b_z = torch.full(shp_gf, 5.0, device="cuda"), torch.full(shp_z, 5.0, device="cuda")
            pa, pb = problems(a_gf, a_z, lives_now), problems(b_gf, b_z, lives_now)
            assert ops.linear_bwd_weight_gathered_multi_ok(F, pa)
            ops.linear_bwd_weight_gathered_multi(Xp, F, pa, epoch=epoch)
            for q in pb:
                ops.linear_bwd_weight_gathered(q["dh"], Xp, F, q["ids"], q["dw"], q["ind_code"], epoch, q["num_ind"], d_n=q["d_n"],
                                               accumulate=q["accumulate"], ind_mask=q["ind_mask"], split=True)
            torch.cuda.synchronize()
            ref_gf = torch.zeros((fo, F + ni), dtype=torch.float64); mag_gf = torch.zeros_like(ref_gf)
            ref_z = torch.zeros((fo, F), dtype=torch.float64); mag_z = torch.zeros_like(ref_z)
            for q, ((ids, dh, _), n) in enumerate(zip(probs_np, lives_now)):
                if n == 0:
                    continue
                feat = torch.from_numpy(X_np[ids[:n]]).double()
                if q != 2:          # the indicator columns this hop's forward pass saw: bit j of the node's word under the hop's mask
                    bits = code_np[ids[:n]] & 0xff & (3 if q == 0 else 7)
                    feat = torch.cat([feat, torch.from_numpy(((bits[:, None] >> np.arange(ni)) & 1).astype(np.float64))], 1)
                d = torch.from_numpy(dh[:n]).double()
                if q == 2:
                    ref_z += d.t() @ feat; mag_z += d.abs().t() @ feat.abs()
                else:
                    ref_gf += d.t() @ feat; mag_gf += d.abs().t() @ feat.abs()
            for name, got_a, got_b, ref, mag in (("gf", a_gf, b_gf, ref_gf, mag_gf), ("z", a_z, b_z, ref_z, mag_z)):
                K = ref.shape[1]
                assert bool(torch.isfinite(got_a).all())
                if got_a.shape[1] > K:
                    assert float(got_a[:, K:].abs().sum()) == 0.0
                mg = mag.clamp_min(1e-30)
                ea = float(((got_a[:, :K].cpu().double() - ref).abs() / mg).max())
                eb = float(((got_b[:, :K].cpu().double() - ref).abs() / mg).max())
                assert ea < 1e-6 and ea <= 1.1 * eb + 2e-8, (name, layout, lives_now, ea, eb)


def test_adam_with_pending_slab_sums_when_no_row_is_live():
    """grapes_adam_step_slabs with a live row count of 0 (no slab was written) must behave like
    grapes_slab_reduce_sets followed by grapes_adam_step — a zero gradient (or the untouched one when accumulating) — and must
    not read the slab in front of the workspace."""
    _cuda()
    from grapes_amd import ops
    n, fi, fo = 300, 64, 32
    rng = np.random.default_rng(5)
    for d_live in (0, 37):
        res = []
        for fused_slabs in (True, False):
            torch.manual_seed(1)
            w = torch.nn.Parameter(torch.randn(fo, fi, device="cuda")); b = torch.nn.Parameter(torch.randn(fo, device="cuda"))
            w.grad = torch.full_like(w, 0.25); b.grad = torch.full_like(b, -0.5)
            opt = torch.optim.Adam([w, b], lr=1e-2, capturable=True)
            fa = ops.FusedAdam([opt])
            dout = _t(rng.standard_normal((n, fo)).astype(np.float32)); x = _t(rng.standard_normal((n, fi)).astype(np.float32))
            d_n = torch.tensor([d_live], dtype=torch.int32, device="cuda")
            ds = ops.DeferredSlabs()
            assert ds.try_add(dout, None, x, d_n, w.grad, b.grad, False)
            if fused_slabs:
                fa.step(slabs=ds)                  # the sums happen inside the update launch
                assert not ds.sets
            else:
                ds.flush()                         # grapes_slab_reduce_sets, then the plain update
                fa.step()
            torch.cuda.synchronize()
            res.append((w.detach().clone(), b.detach().clone(), w.grad.clone(), b.grad.clone()))
            rng = np.random.default_rng(5)         # the same operands for the second form
        for u, v in zip(*res):
            assert torch.equal(u, v), d_live
        if d_live == 0:
            assert float(res[0][2].abs().max()) == 0.0 and float(res[0][3].abs().max()) == 0.0


def test_reference_shaped_loop_over_the_drop_in_modules_matches_the_oracle():
    """the reference's loop (main.py:157-291: host masks, CPU data.x, per-hop H2D, int64 CPU indices) with
    only its three import lines changed — grapes_amd.reference_loop — against the oracle on the same batches and uniforms:
    kept sets and all_nodes bit-exact, logits 1e-5, losses; three steps with both optimisers."""
    _cuda()
    from grapes_amd import synth
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.reference_loop import ReferenceShapedLoop
    from oracle import grapes_oracle as O
    n, deg, F, C, B, K, hops, H = 20000, 10.0, 50, 6, 96, 64, 2, 128
    indptr, indices = synth.synth_csr_numpy(n, deg, 1500, seed=2)
    rng = np.random.default_rng(3)
    X = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, C, n))
    torch.manual_seed(0)
    rc, rgf, rz = O.GCNRef(F, [H, C]), O.GCNRef(F + hops + 1, [H, 1]), O.GCNRef(F, [H, 1])
    c, gf, z = GCN(F, [H, C]).cuda(), GCN(F + hops + 1, [H, 1]).cuda(), GCN(F, [H, 1]).cuda()
    c.load_state_dict(rc.state_dict()); gf.load_state_dict(rgf.state_dict()); z.load_state_dict(rz.state_dict())
    oc = torch.optim.Adam(c.parameters(), lr=1e-3); og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=1e-4)
    roc = torch.optim.Adam(rc.parameters(), lr=1e-3); rog = torch.optim.Adam(list(rgf.parameters()) + list(rz.parameters()), lr=1e-4)
    loop = ReferenceShapedLoop(DeviceGraph.from_csr(indptr, indices), X, y, c, gf, z, sampling_hops=hops, num_samples=K,
                               loss_coef=30.0, optimizer_c=oc, optimizer_gf=og)
    node_map = O.TensorMap(n)
    perm = rng.permutation(n)
    for s in range(3):
        tg = perm[s * B:(s + 1) * B].astype(np.int64)
        uni = {h: rng.random(n, dtype=np.float32) for h in range(hops)}
        out = loop.step(torch.from_numpy(tg), uniforms_fn=lambda h, nn: torch.from_numpy(uni[h][:nn]).cuda())
        ot = O.train_step(indptr, indices, X, y, tg, rc, rgf, rz, sampling_hops=hops, num_samples=K,
                          uniforms_fn=lambda h, nn: uni[h][:nn], loss_coef=30.0, optimizer_c=roc, optimizer_gf=rog, node_map=node_map)
        for hop in range(hops):
            assert np.array_equal(out["kept"][hop].numpy(), ot["hops"][hop]["kept"]), (s, hop)
        assert np.array_equal(out["all_nodes"].numpy(), ot["all_nodes"]), s
        ref = ot["logits"].numpy()
        tol = 1e-5 if s == 0 else 2e-4
        assert float(np.abs(out["logits"].cpu().numpy() - ref).max()) <= tol * max(1.0, float(np.abs(ref).max())), s
        assert abs(out["loss_c"] - ot["loss_c"]) <= tol * max(1.0, abs(ot["loss_c"])), s
        assert abs(out["loss_gfn"] - ot["loss_gfn"]) <= 10 * tol * max(1.0, abs(ot["loss_gfn"])), s
