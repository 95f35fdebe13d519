"""csrc/exchange_kernels.hip on ONE MI355X with P simulated ranks: every rank's shard lives on the same GPU, the
all-gather / all-to-all are done by hand (concatenate / transpose the slots), and the kernels either side of them
must reproduce (a) the oracle's get_neighborhoods (utils.py:74-82) / X[ids] bit for bit and (b) the messages of the
CPU double that the gloo tests run (tests/dist_worker.OracleLocalOps), so the two test families pin the same layout."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _graph(N, seed):
    from oracle import grapes_oracle as O
    rng = np.random.default_rng(seed)
    ei = rng.integers(0, N, (2, 60000))
    ei[0, :5000] = 17                                     # a hub
    ei[0, 5000:5200] = N - 1
    indptr, indices = O.build_csr(np.concatenate([ei, ei[::-1]], axis=1), N)
    return indptr, indices


@pytest.mark.parametrize("P,F,num_ind", [(1, 8, 0), (3, 100, 4), (8, 7, 3)])
def test_exchange_kernels_simulated_ranks(P, F, num_ind):
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from dist_worker import OracleLocalOps
    from grapes_amd import ops
    from grapes_amd.dist import partition_bounds
    from oracle import grapes_oracle as O
    dev = torch.device("cuda", 0)
    N = 5003
    indptr, indices = _graph(N, 3)
    rng = np.random.default_rng(4)
    X = rng.standard_normal((N, F)).astype(np.float32)
    b = partition_bounds(N, P)
    bounds32 = torch.tensor(b, dtype=torch.int32, device=dev)
    rp_full, col_full = torch.from_numpy(indptr), torch.from_numpy(indices.astype(np.int32))
    shards = []
    for r in range(P):
        lo, hi = b[r], b[r + 1]
        rp = (rp_full[lo:hi + 1] - rp_full[lo]).clone()
        cl = col_full[int(rp_full[lo]):int(rp_full[hi])].clone()
        shards.append((rp, cl, torch.from_numpy(X[lo:hi].copy())))
    dbl = OracleLocalOps()

    # ---------------- adjacency rows
    cap, e_cap = 300, 40000
    stride = 2 * cap + e_cap
    queries, counts = [], []
    for r in range(P):
        q = rng.permutation(N)[:cap].astype(np.int32)
        q[:4] = [17, N - 1, 17, 0]                                          # hub twice, last node, first node
        queries.append(q); counts.append(cap - 11 * r)
    req_cpu = torch.cat([torch.cat([torch.from_numpy(q), torch.tensor([m], dtype=torch.int32)])
                         for q, m in zip(queries, counts)])
    req = req_cpu.to(dev)
    replies, replies_cpu = [], []
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    for o in range(P):
        rp, cl, _ = shards[o]
        reply = torch.zeros(P * stride, dtype=torch.int32, device=dev)
        ops.exchange_serve_rows(rp.to(dev), cl.to(dev), req, P, cap, b[o], b[o + 1], reply, stride, e_cap, status=status)
        ref = torch.zeros(P * stride, dtype=torch.int32)
        dbl.serve_rows(rp, cl, req_cpu, P, cap, b[o], b[o + 1], ref, stride, e_cap, torch.zeros(1, dtype=torch.int32))
        assert torch.equal(reply.cpu(), ref), f"owner {o}: reply slots differ from the CPU double"
        replies.append(reply.view(P, stride)); replies_cpu.append(ref.view(P, stride))
    assert int(status) == 0
    for r in range(P):
        back = torch.stack([replies[o][r] for o in range(P)]).reshape(-1).contiguous()      # the all-to-all, by hand
        nodes = torch.from_numpy(queries[r]).to(dev)
        d_m = torch.tensor([counts[r]], dtype=torch.int32, device=dev)
        src, dst, d_e, eoff = ops.exchange_recv_rows(back, stride, nodes, bounds32, P, e_cap, d_m=d_m, status=status)
        ref = O.get_neighborhoods(queries[r][:counts[r]].astype(np.int64), indptr, indices)
        e = ref.shape[1]
        assert int(d_e) == e and int(eoff[counts[r]]) == e
        assert np.array_equal(src[:e].cpu().numpy().astype(np.int64), ref[0])
        assert np.array_equal(dst[:e].cpu().numpy().astype(np.int64), ref[1])
    assert int(status) == 0
    # a requester whose total exceeds e_cap raises the status word and stays inside its buffers
    small = 1000
    back = torch.stack([replies[o][0] for o in range(P)]).reshape(-1).contiguous()
    st2 = torch.zeros(1, dtype=torch.int32, device=dev)
    src, dst, d_e, _ = ops.exchange_recv_rows(back, stride, torch.from_numpy(queries[0]).to(dev), bounds32, P, small,
                                              d_m=None, status=st2)
    assert int(st2) & 1 and src.numel() == small

    # ---------------- halo feature rows
    capf = 1200
    lists, ns = [], []
    for r in range(P):
        ids = np.sort(rng.permutation(N)[:capf]).astype(np.int32)
        lists.append(ids); ns.append(capf - 37 * r)
    reqf_cpu = torch.cat([torch.cat([torch.from_numpy(i), torch.tensor([n], dtype=torch.int32)])
                          for i, n in zip(lists, ns)])
    reqf = reqf_cpu.to(dev)
    n_slot = capf if P == 1 else 2 * capf // P
    freplies = []
    for o in range(P):
        reply = torch.zeros(P * n_slot * F, dtype=torch.float32, device=dev)
        ops.exchange_serve_features(shards[o][2].to(dev), reqf, P, capf, b[o], b[o + 1], reply, n_slot, status=status)
        ref = torch.zeros(P * n_slot * F, dtype=torch.float32)
        dbl.serve_features(shards[o][2], reqf_cpu, P, capf, b[o], b[o + 1], ref, n_slot, torch.zeros(1, dtype=torch.int32))
        assert torch.equal(reply.cpu(), ref)
        freplies.append(reply.view(P, n_slot * F))
    assert int(status) == 0
    epoch = 5
    code = torch.from_numpy(((epoch << 8) | rng.integers(0, 1 << max(num_ind, 1), N)).astype(np.int32))
    code[::3] = (4 << 8) | 0xf                                              # stale epoch: indicators read as 0
    for r in range(P):
        back = torch.stack([freplies[o][r] for o in range(P)]).reshape(-1).contiguous()
        ids = torch.from_numpy(lists[r]).to(dev)
        d_n = torch.tensor([ns[r]], dtype=torch.int32, device=dev)
        out = ops.exchange_assemble_features(back, F, n_slot, ids, bounds32, P, d_n=d_n, ind_code=code.to(dev) if num_ind else None,
                                             epoch=epoch, num_ind=num_ind)
        n = ns[r]
        assert out.shape == (capf, F + num_ind)
        assert np.array_equal(out[:n, :F].cpu().numpy(), X[lists[r][:n]])                  # halo rows are bit copies
        if num_ind:
            c = code[torch.from_numpy(lists[r][:n]).long()].numpy()
            want = np.stack([((c >> j) & 1) * ((c >> 8) == epoch) for j in range(num_ind)], axis=1).astype(np.float32)
            assert np.array_equal(out[:n, F:].cpu().numpy(), want)
        ref = dbl.assemble_features(back.cpu(), F, n_slot, ids.cpu(), bounds32.cpu(), P, d_n.cpu(), code if num_ind else None,
                                    epoch, None, num_ind)
        assert torch.equal(out[:n].cpu(), ref[:n])
    # a halo slot that is too small raises NODE_OVERFLOW
    st3 = torch.zeros(1, dtype=torch.int32, device=dev)
    tiny = torch.zeros(P * 8 * F, dtype=torch.float32, device=dev)
    ops.exchange_serve_features(shards[0][2].to(dev), reqf, P, capf, b[0], b[1], tiny, 8, status=st3)
    assert int(st3) & 2


def test_pack_query_message():
    """[cap ids | live count] in one launch: ids copied, padding untouched, count clamped to the list length."""
    from grapes_amd import ops
    ids = torch.arange(100, 700, dtype=torch.int32, device="cuda")
    for cap, d_n, want in ((600, None, 600), (900, torch.tensor([123], dtype=torch.int32, device="cuda"), 123),
                           (900, torch.tensor([5000], dtype=torch.int32, device="cuda"), 600)):
        q = torch.full((cap + 1,), -7, dtype=torch.int32, device="cuda")
        ops.exchange_pack_query(ids, d_n, cap, q)
        assert torch.equal(q[:600], ids) and int(q[cap]) == want and bool((q[600:cap] == -7).all())
