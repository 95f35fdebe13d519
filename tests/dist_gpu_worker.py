"""Rank program of tests/test_dist_gpu.py::test_two_process_partition_on_one_gpu — TWO real processes share GPU 0 and run
the 1-D partitioned captured step with the HIP exchange kernels (pack / serve / receive / assemble, csrc/exchange_kernels.hip)
on both sides of every collective.  RCCL refuses two ranks on one device, so the transport is gloo with the message buffers
staged through the host; everything else — kernels, buffers, capacities, hipGraph segments — is the production path.
Each rank checks its step against the single-GPU captured step on the same batch: sampled sets and activations do not
depend on P (halo rows are bit copies), and the all-reduced gradients equal the mean of the two ranks' local gradients."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    from grapes_amd import synth
    from grapes_amd.dist import GradSync, shard_full_graph
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.modules.gcn import GCN
    from grapes_amd.step_graph import GraphedTrainer

    def staged(graph):                                     # gloo transport through the host for the persistent device buffers
        def ag(out, inp):
            o = torch.empty(out.shape, dtype=out.dtype)
            graph.run_collective(lambda: (dist.all_gather_into_tensor(o, inp.cpu()), out.copy_(o)))
            graph.exchanged_bytes += inp.numel() * inp.element_size() * max(world - 1, 1)
        def a2a(out, inp):
            o = torch.empty(out.shape, dtype=out.dtype)
            graph.run_collective(lambda: (dist.all_to_all_single(o, inp.cpu()), out.copy_(o)))
            graph.exchanged_bytes += inp.numel() * inp.element_size()
        graph._all_gather, graph._all_to_all = ag, a2a
        return graph

    class StagedGradSync(GradSync):
        def _all_reduce(self, flat):
            t = flat.cpu(); dist.all_reduce(t); flat.copy_(t)

    n, deg, F, C, B, K, hops, H = 20000, 11.0, 100, 9, 96, 64, 3, 256      # F + hops + 1 = 104: the in-place halo path (no assembled copy)
    indptr, indices = synth.synth_csr_numpy(n, deg, 1500, seed=21)
    rng = np.random.default_rng(22)
    X = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.integers(0, C, n)).cuda()
    train = torch.from_numpy(rng.permutation(n)[:1500].astype(np.int64)).cuda()
    rowptr, col = torch.from_numpy(indptr).cuda(), torch.from_numpy(indices).cuda()
    maxd = int((rowptr[1:] - rowptr[:-1]).max())

    def models():
        torch.manual_seed(0)
        return GCN(F, [H, H, C]).cuda(), GCN(F + hops + 1, [H, 1]).cuda(), GCN(F, [H, 1]).cuda()

    def run(kind, stripe):
        c, gf, z = models()
        if kind == "single":
            g, Xa, gs = DeviceGraph(rowptr, col, n), X, None
        elif kind == "peers":      # the other rank's shard is read IN PLACE through its hipIpc mapping: no exchange, one collective
            g, Xa, gs = DeviceGraph(rowptr, col, n), peer_x, StagedGradSync(world)
        else:
            g = staged(shard_full_graph(rowptr, col, X, rank, world, max_degree=maxd, replicate_adjacency=(kind == "repl_adj")))
            Xa, gs = None, StagedGradSync(world)
        tr = GraphedTrainer(g, Xa, y, c, gf, z, batch_size=B, sampling_hops=hops, num_samples=K, loss_coef=50.0,
                            e_cap=1 << 15, philox_seed=7 + stripe, capture=True, grad_sync=gs)
        tr.attach_loader(train, stride=world, offset=stripe)
        outs = []
        for s in range(5):                                  # eager warm-up steps, capture, replays
            o = tr.step_next()
            torch.cuda.synchronize()
            tr.check()
            outs.append(dict(kept=[k.clone() for k in o["kept"]], kc=[int(x) for x in o["kept_counts"]], na=int(o["n_all"]),
                             alln=o["all_nodes"].clone(), logits=o["logits"].clone(), loss_c=float(o["loss_c"]),
                             loss_gfn=float(o["loss_gfn"]),
                             grads=[p.grad.clone() for m in (c, gf, z) for p in m.parameters()]))
        return outs, tr

    # every rank keeps ONLY its own rows in the shard it exports; what it reads of the other rows comes through the mapping
    from grapes_amd.dist import partition_bounds
    from grapes_amd.peer import PeerFeatures
    pb = partition_bounds(n, world)
    peer_x = PeerFeatures.open(X[pb[rank]:pb[rank + 1]].clone(), pb, rank, world)
    assert len(peer_x._opened) == world - 1
    probe = torch.from_numpy(rng.integers(0, n, 64))
    assert torch.equal(peer_x.rows_for_check(probe), X[probe.cuda()])
    ref = {st: run("single", st)[0] for st in range(world)}   # both stripes on one GPU, no exchange, no gradient sync
    for kind in ("peers", "part_adj", "repl_adj"):
        outs, tr = run(kind, rank)
        assert tr.graph_obj is not None
        if kind == "peers":
            assert tr.graph_obj.num_collectives == 1 and tr.graph_obj.num_segments == 2
        else:
            assert tr.g.exchanged_bytes > 0
        for s, (o, r) in enumerate(zip(outs, ref[rank])):
            assert o["kc"] == r["kc"] and o["na"] == r["na"], (kind, s)
            for hop in range(hops):
                assert torch.equal(o["kept"][hop][:o["kc"][hop]], r["kept"][hop][:r["kc"][hop]]), (kind, s, hop)
            assert torch.equal(o["alln"][:o["na"]], r["alln"][:r["na"]]), (kind, s)
            assert torch.equal(o["logits"][:o["na"]], r["logits"][:r["na"]]), (kind, s)     # halo rows are bit copies
            assert o["loss_c"] == r["loss_c"] and o["loss_gfn"] == r["loss_gfn"], (kind, s)
            for i, gsync in enumerate(o["grads"]):           # all-reduced mean of the ranks' local gradients
                mean = sum(ref[st][s]["grads"][i] for st in range(world)) / world
                scale = max(1.0, float(mean.abs().max()))
                assert float((gsync - mean).abs().max()) <= 1e-5 * scale, (kind, s, i)
        print(f"rank {rank}/{world} {kind} ok: {tr.graph_obj.num_segments} segments, {tr.graph_obj.num_collectives} collectives", flush=True)
    dist.barrier()
    peer_x.close()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok", flush=True)


if __name__ == "__main__":
    main()
