"""Generates tests/golden/*.npz by importing the REFERENCE's own modules/utils.py.

Run only in the build container (needs /root/reference; it never travels to the GPU box):

    python tests/golden/make_golden.py

The fixtures are data only (inputs + the reference's outputs).  SciPy here is 1.15.3, not the
pinned 1.13.1, and rejects torch tensors as CSR fancy indices; the reference code is left
untouched and is handed a csr_matrix *subclass* that converts tensor keys to numpy
(SURVEY.md §8c caveat).
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
import modules.utils as RU  # noqa: E402  (the reference)

OUT = os.path.dirname(os.path.abspath(__file__))


class TCSR(sp.csr_matrix):
    def __getitem__(self, key):
        conv = lambda k: k.numpy() if isinstance(k, torch.Tensor) else k
        key = tuple(conv(k) for k in key) if isinstance(key, tuple) else conv(key)
        r = sp.csr_matrix.__getitem__(self, key)
        return TCSR(r) if sp.issparse(r) else r


def rand_graph(rng, n, e, symmetric=True, self_loops=True, dups=True):
    src = rng.integers(0, n, e)
    dst = rng.integers(0, n, e)
    if dups:  # duplicate pairs pin the constructor's dedup (main.py:134-136)
        src = np.concatenate([src, src[: e // 5]])
        dst = np.concatenate([dst, dst[: e // 5]])
    if self_loops:
        loops = rng.integers(0, n, max(1, n // 10))
        src = np.concatenate([src, loops])
        dst = np.concatenate([dst, loops])
    if symmetric:
        src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    return np.stack([src, dst]).astype(np.int64)


def make_adj(edge_index, n):
    # exactly main.py:134-136
    return TCSR(sp.csr_matrix((np.ones(edge_index.shape[1], dtype=bool), edge_index), shape=(n, n)))


def g1_g2():
    rng = np.random.default_rng(11)
    out = {}
    for tag, (n, e, sym) in {"a": (40, 150, True), "b": (300, 2500, False), "c": (1000, 30000, True)}.items():
        ei = rand_graph(rng, n, e, symmetric=sym)
        A = make_adj(ei, n)
        out[f"{tag}_edge_index"] = ei
        out[f"{tag}_n"] = np.int64(n)
        out[f"{tag}_indptr"] = A.indptr.astype(np.int64)
        out[f"{tag}_indices"] = A.indices.astype(np.int64)
        # G1
        nodes = torch.from_numpy(rng.permutation(n)[: max(3, n // 7)].astype(np.int64))
        out[f"{tag}_nodes"] = nodes.numpy()
        out[f"{tag}_neigh"] = RU.get_neighborhoods(nodes, A).numpy()
        nodes_dup = torch.cat([nodes[:5], nodes[:3]])  # repeated query nodes
        out[f"{tag}_nodes_dup"] = nodes_dup.numpy()
        out[f"{tag}_neigh_dup"] = RU.get_neighborhoods(nodes_dup, A).numpy()
        # G2, both argument orders (main.py:241-243, eval.py:140-142)
        rows = torch.from_numpy(rng.permutation(n)[: max(4, n // 5)].astype(np.int64))
        cols = torch.from_numpy(rng.permutation(n)[: max(4, n // 4)].astype(np.int64))
        out[f"{tag}_rows"], out[f"{tag}_cols"] = rows.numpy(), cols.numpy()
        out[f"{tag}_slice_rc"] = RU.slice_adjacency(A, rows, cols).numpy()
        out[f"{tag}_slice_cr"] = RU.slice_adjacency(A, cols, rows).numpy()
        cols_dup = torch.cat([cols, cols[:4]])
        out[f"{tag}_cols_dup"] = cols_dup.numpy()
        out[f"{tag}_slice_dup"] = RU.slice_adjacency(A, rows, cols_dup).numpy()
        empty = torch.zeros(0, dtype=torch.long)
        out[f"{tag}_slice_emptycols"] = RU.slice_adjacency(A, rows, empty).numpy()
    np.savez_compressed(os.path.join(OUT, "g1_g2_csr.npz"), **out)


def g3():
    out = {}
    nodes = torch.tensor([22, 32, 42, 52])                      # utils.py:103-108 docstring vector
    tm = RU.TensorMap(size=int(nodes.max()) + 1)
    tm.update(nodes)
    q = torch.tensor([52, 42, 32, 22, 22])
    out["doc_keys"], out["doc_query"], out["doc_out"] = nodes.numpy(), q.numpy(), tm.map(q).numpy()
    # stale entries persist between updates
    tm2 = RU.TensorMap(size=64)
    k1 = torch.tensor([5, 9, 60, 1, 33])
    k2 = torch.tensor([9, 2, 40])
    tm2.update(k1)
    r1 = tm2.map(k1).numpy()
    tm2.update(k2)
    qq = torch.tensor([5, 9, 60, 1, 33, 2, 40])
    out["seq_k1"], out["seq_k2"], out["seq_r1"] = k1.numpy(), k2.numpy(), r1
    out["seq_query"], out["seq_out"] = qq.numpy(), tm2.map(qq).numpy()
    np.savez_compressed(os.path.join(OUT, "g3_tensormap.npz"), **out)


def run_ref_sampler(logits, nodes, k, seed):
    """Calls the reference sampler; returns its outputs plus the uniforms its Gumbel draw used."""
    n = nodes.shape[0]
    torch.manual_seed(seed)
    r = torch.rand(n)
    torch.manual_seed(seed)
    kept, logp, stats = RU.sample_neighborhoods_from_probs(logits.clone(), nodes, k)
    # reproduce the reference's keys from r with torch ops to make sure r IS the draw (utils.py:40-42)
    fi = torch.finfo(torch.float32)
    u = fi.tiny + r * ((1 - fi.eps) - fi.tiny)
    g = -torch.log(-torch.log(u))
    torch.manual_seed(seed)
    g_ref = torch.distributions.Gumbel(torch.tensor(0.0), torch.tensor(1.0)).sample((n,))
    assert torch.equal(g, g_ref), "uniforms do not reproduce the reference's Gumbel draw"
    keys = torch.sigmoid(logits.squeeze()).log() + g
    return r, kept, logp, stats, keys


def g4():
    out = {}
    cases = [(3, 5), (12, 4), (12, 11), (12, 12), (4096, 256), (40000, 256), (40000, 512)]
    rng = np.random.default_rng(4)
    names = []
    for ci, (n, k) in enumerate(cases):
        for seed in (0, 1):
            tag = f"n{n}_k{k}_s{seed}"
            names.append(tag)
            torch.manual_seed(1000 + 17 * ci + seed)
            logits = (torch.randn(n, 1) * 3.0)
            nodes = torch.from_numpy(np.sort(rng.permutation(10 * n + 7)[:n]).astype(np.int64))
            out[f"{tag}_logits"], out[f"{tag}_nodes"], out[f"{tag}_k"] = logits.numpy(), nodes.numpy(), np.int64(k)
            if k >= n:
                kept, logp, stats = RU.sample_neighborhoods_from_probs(logits.clone(), nodes, k)
                assert stats == {}
                out[f"{tag}_kept"], out[f"{tag}_logp"] = kept.numpy(), logp.numpy()
                continue
            r, kept, logp, stats, keys = run_ref_sampler(logits, nodes, k, seed)
            out[f"{tag}_uniforms"], out[f"{tag}_kept"], out[f"{tag}_logp"] = r.numpy(), kept.numpy(), logp.numpy()
            out[f"{tag}_keys"] = keys.numpy()
            out[f"{tag}_stats"] = np.array([float(stats[s]) for s in
                                            ("min_prob", "max_prob", "mean_entropy", "std_entropy")], dtype=np.float64)
    # extreme logits: log(sigmoid) underflow -> -inf keys, NaN entropy -> 0, denormal log-probs
    ext = torch.tensor([50.0, -50.0, 120.0, -120.0, 200.0, -200.0, 100.0, -100.0, 0.0, 88.0, -88.0, -87.0,
                        3.0, -3.0, 1.5, -0.5]).reshape(-1, 1)
    nodes = torch.arange(100, 100 + ext.shape[0])
    for k in (4, 12):
        tag = f"extreme_k{k}"
        names.append(tag)
        r, kept, logp, stats, keys = run_ref_sampler(ext, nodes, k, 7)
        out[f"{tag}_logits"], out[f"{tag}_nodes"], out[f"{tag}_k"] = ext.numpy(), nodes.numpy(), np.int64(k)
        out[f"{tag}_uniforms"], out[f"{tag}_kept"], out[f"{tag}_logp"] = r.numpy(), kept.numpy(), logp.numpy()
        out[f"{tag}_keys"] = keys.numpy()
        out[f"{tag}_stats"] = np.array([float(stats[s]) for s in
                                        ("min_prob", "max_prob", "mean_entropy", "std_entropy")], dtype=np.float64)
    # random_sampling: constant logits 100 (main.py:207)
    n, k = 500, 64
    tag = "const100"
    names.append(tag)
    logits = 100 * torch.ones((n, 1))
    nodes = torch.arange(n) * 3
    r, kept, logp, stats, keys = run_ref_sampler(logits, nodes, k, 3)
    out[f"{tag}_logits"], out[f"{tag}_nodes"], out[f"{tag}_k"] = logits.numpy(), nodes.numpy(), np.int64(k)
    out[f"{tag}_uniforms"], out[f"{tag}_kept"], out[f"{tag}_logp"] = r.numpy(), kept.numpy(), logp.numpy()
    out[f"{tag}_keys"] = keys.numpy()
    out[f"{tag}_stats"] = np.array([float(stats[s]) for s in
                                    ("min_prob", "max_prob", "mean_entropy", "std_entropy")], dtype=np.float64)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "g4_sampler.npz"), **out)


def g6():
    """Near-tie family: the build computes the Gumbel-top-k keys with a portable exp / log that is
    bit-identical on CPU and GPU but differs from torch's by up to ~2e-6, so a k-th / (k+1)-th key gap below that can select
    a different node than the reference.  (a) natural draws: how small the gap gets by itself; (b) engineered draws: the
    logit of the (k+1)-th candidate is moved so that its key lands a chosen distance below the k-th.  Inputs (logits,
    uniforms), the reference's kept set and its float32 key gap are stored; tests/test_oracle_golden.py measures where the
    portable keys start to disagree."""
    import math
    out, names = {}, []
    rng = np.random.default_rng(66)
    n, k = 2048, 128
    nodes = torch.arange(n, dtype=torch.int64) * 5 + 1

    def record(tag, logits, seed):
        r, kept, logp, stats, keys = run_ref_sampler(logits, nodes, k, seed)
        ks, _ = torch.sort(keys, descending=True)
        out[f"{tag}_logits"], out[f"{tag}_uniforms"] = logits.numpy().astype(np.float32), r.numpy()
        out[f"{tag}_kept"], out[f"{tag}_gap"] = kept.numpy(), np.float64(float(ks[k - 1]) - float(ks[k]))
        names.append(tag)
        return r, keys

    for d in range(48):                                                    # (a) natural draws
        torch.manual_seed(6000 + d)
        record(f"nat{d}", torch.randn(n, 1) * 2.5, 300 + d)
    for ci, delta in enumerate([1e-3, 1e-4, 3e-5, 1e-5, 6e-6, 4e-6, 3e-6, 2e-6, 1.5e-6, 1e-6, 7e-7, 5e-7, 3e-7, 2e-7, 1e-7, 0.0]):
        for rep in range(3):                                                # (b) engineered gaps
            seed = 900 + 10 * ci + rep
            torch.manual_seed(7000 + 10 * ci + rep)
            logits = torch.randn(n, 1) * 2.5
            torch.manual_seed(seed)
            r = torch.rand(n)
            fi = torch.finfo(torch.float32)
            g = -torch.log(-torch.log(fi.tiny + r * ((1 - fi.eps) - fi.tiny)))
            keys = torch.sigmoid(logits.squeeze()).log() + g
            order = torch.argsort(keys, descending=True)
            j, kth = int(order[k]), float(keys[order[k - 1]])
            t = kth - delta - float(g[j])                                   # wanted log sigmoid(l_j)
            if t >= -1e-6:
                continue
            logits[j, 0] = t - math.log1p(-math.exp(t))                     # inverse of logsigmoid (float64)
            record(f"eng{ci}_{rep}", logits, seed)
    out["names"] = np.array(names)
    out["n"], out["k"] = np.int64(n), np.int64(k)
    out["nodes"] = nodes.numpy()
    np.savez_compressed(os.path.join(OUT, "g6_near_ties.npz"), **out)


def injected_logits(hop, batch_nodes):
    v = batch_nodes.to(torch.float64)
    return (3.0 * torch.sin(0.37 * v + 1.3 * hop)).to(torch.float32).reshape(-1, 1)


def g5():
    """Index pipeline of one training iteration (main.py:157-256) driven through the reference's
    functions with *injected* logits, so that no GCN arithmetic is involved."""
    rng = np.random.default_rng(5)
    out = {}
    for tag, (n, e, B, K, hops) in {"small": (200, 900, 16, 8, 2), "mid": (3000, 30000, 64, 32, 3)}.items():
        ei = rand_graph(rng, n, e, symmetric=True)
        A = make_adj(ei, n)
        out[f"{tag}_indptr"], out[f"{tag}_indices"] = A.indptr.astype(np.int64), A.indices.astype(np.int64)
        out[f"{tag}_cfg"] = np.array([n, B, K, hops], dtype=np.int64)
        target = torch.from_numpy(rng.permutation(n)[:B].astype(np.int64))
        out[f"{tag}_targets"] = target.numpy()
        node_map = RU.TensorMap(size=n)
        prev_mask = torch.zeros(n, dtype=torch.bool)
        batch_mask = torch.zeros(n, dtype=torch.bool)
        ind = torch.zeros((n, hops + 1))
        previous = target.clone()
        all_mask = torch.zeros(n, dtype=torch.bool)
        all_mask[target] = True
        ind[target, -1] = 1.0
        g_edges = []
        for hop in range(hops):
            nb = RU.get_neighborhoods(previous, A)
            prev_mask.zero_(); batch_mask.zero_()
            prev_mask[previous] = True
            batch_mask[nb.view(-1)] = True
            nbm = batch_mask & ~prev_mask
            batch_nodes = node_map.values[batch_mask]
            neighbor_nodes = node_map.values[nbm]
            ind[neighbor_nodes, hop] = 1.0
            node_map.update(batch_nodes)
            local = node_map.map(nb)
            logits = injected_logits(hop, batch_nodes)[node_map.map(neighbor_nodes)]
            seed = 50 + hop
            torch.manual_seed(seed)
            r = torch.rand(neighbor_nodes.shape[0])
            torch.manual_seed(seed)
            kept, logp, _ = RU.sample_neighborhoods_from_probs(logits, neighbor_nodes, K)
            all_mask[kept] = True
            nxt = torch.cat([target, kept])
            khe = RU.slice_adjacency(A, rows=nxt, cols=previous)
            g_edges.append(khe)
            p = f"{tag}_h{hop}_"
            out[p + "neigh"], out[p + "batch_nodes"], out[p + "neighbor_nodes"] = nb.numpy(), batch_nodes.numpy(), neighbor_nodes.numpy()
            out[p + "local"], out[p + "ind_rows"] = local.numpy(), ind[batch_nodes].numpy()
            out[p + "uniforms"], out[p + "kept"], out[p + "logp"] = r.numpy(), kept.numpy(), logp.numpy()
            out[p + "k_hop_edges"] = khe.numpy()
            previous = nxt.clone()
        all_nodes = node_map.values[all_mask]
        node_map.update(all_nodes)
        out[f"{tag}_all_nodes"] = all_nodes.numpy()
        for i, ed in enumerate(g_edges):
            out[f"{tag}_edge_index_{i}"] = node_map.map(ed).numpy()
        out[f"{tag}_local_targets"] = node_map.map(target).numpy()
    np.savez_compressed(os.path.join(OUT, "g5_step_trace.npz"), **out)


if __name__ == "__main__":
    g1_g2(); g3(); g4(); g5(); g6()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
