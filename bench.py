#!/usr/bin/env python3
"""bench.py — GRAPES training-step throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one full training iteration of reference main.py:157-291 (3 sampling hops with the
GFlowNet sampler GCN + exact-k Gumbel-top-k draw, log-Z net, classifier fwd/bwd, Trajectory-Balance
loss, both Adam steps) on one mini-batch of B target nodes of a synthetic ogbn-products-shaped
graph resident in HBM.  value = GCNConv edge aggregations per second, whole job.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline"     — the gather-SpMM (gcn_aggregate) kernel: algorithmic bytes / HIP-event time vs 8 TB/s
  "cpu_baseline" — the CPU oracle (port of the reference control flow with torch-CPU GCNConv) timed on
                   this box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
DEFAULT_DISPATCH_RAMP_US = 1.6   # rocprofv3 kernel-trace duration minus in-kernel stamps (profiles/bench_static.json: 1.58-1.63)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=1000,
                    help="untimed steps; a fresh process needs about a second of work before clocks and caches settle")
    ap.add_argument("--workload", default="products", choices=["products", "arxiv", "reddit", "cora", "papers100m"])
    ap.add_argument("--hidden_dim", type=int, default=256)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu_steps", type=int, default=32, help="steps of the CPU baseline sample (0 = skip)")
    ap.add_argument("--chain", type=int, default=32,
                    help="captured steps per hipGraphLaunch in the warm-up and the timed loop (GraphedTrainer.run_steps; 1: one launch "
                         "per step).  Steps with collectives between their graph segments are never chained")
    ap.add_argument("--no_roofline", action="store_true")
    ap.add_argument("--no_pipeline", action="store_true",
                    help="A/B: without the prelude pipeline (the next step's weight-independent index chain carried as extra "
                         "workgroups of this step's hop-1 launches)")
    ap.add_argument("--no_median", action="store_true", help="skip the separate event-timed pass (>= 100 extra replays) — counter-collection runs")
    ap.add_argument("--e_cap", type=int, default=0, help="edge capacity per hop expansion of the captured step (0: 2^17, reddit 2^19)")
    ap.add_argument("--partition_adjacency", action="store_true", help="N>1: partition the CSR as well (default: features only)")
    ap.add_argument("--partition_only", action="store_true", help="N>1: skip the replicated-DP measurement beside the partitioned one")
    ap.add_argument("--partition_deadline", type=float, default=240.0,
                    help="N>1: seconds the partitioned phase may take before the replicated-DP number is reported instead")
    ap.add_argument("--halo", choices=["peer", "rccl"], default="peer",
                    help="N>1: how a rank gets the feature rows it does not own — peer: read in place from the owner's HBM over xGMI "
                         "(hipIpc-mapped shards, no exchange; falls back to rccl if the mapping is refused); rccl: all-gather + all-to-all")
    ap.add_argument("--skip_rccl", action="store_true",
                    help="N>1 with --halo peer: do not also measure the RCCL all-gather + all-to-all form of the halo exchange")
    ap.add_argument("--force_peer", action="store_true",
                    help="single process: run the peer-mapped path with X cut into --peer_shards in-process shards (the kernel's shard "
                         "table at work, no link crossed) and a world_size-1 RCCL gradient all-reduce")
    ap.add_argument("--peer_shards", type=int, default=8)
    ap.add_argument("--force_partition", action="store_true",
                    help="single process: run the partitioned (all-to-all) code path through a world_size-1 RCCL group")
    ap.add_argument("--replicate", action="store_true", help="N>1: only the replicated data-parallel step")
    ap.add_argument("--partition", action="store_true", help="(kept for compatibility: the partition is the N>1 default)")
    ap.add_argument("--random_sampling", action="store_true",
                    help="the reference's configs/random/* step (uniform draws, no sampler / log-Z net): NOT the BASELINE headline")
    ap.add_argument("--force_grad_sync", action="store_true",
                    help="single process: run the N>1 replicated code path (gradient all-reduce between graph segments) "
                         "through a world_size-1 RCCL group")
    ap.add_argument("--eager_steps", type=int, default=100,
                    help="N=1: steps of the eager drop-in loop (GrapesTrainer: the reference main.py loop over the drop-in "
                         "modules, one size read-back per hop) timed beside the captured step (0 = skip)")
    ap.add_argument("--launch_timeout", type=float, default=1500.0,
                    help="--gpus N > 1 without WORLD_SIZE: seconds the self-launched ranks may take before they are killed")
    ap.add_argument("--dry_launch", action="store_true",
                    help="exercise only the launcher: the ranks form a gloo group on the CPU, all-reduce one word and rank 0 "
                         "prints {n_gpus, n_ranks_seen}; nothing touches a GPU (tests/test_bench_launch_cpu.py)")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks ourselves — as a CHILD
    process (`python -m torch.distributed.run`, rendezvous on 127.0.0.1), BEFORE anything here has touched the GPU (importing
    torch does not initialise it; nothing is exec'ed or re-exec'ed) — relay the ranks' one JSON line and exit with the
    launcher's code (non-zero when a rank failed; 4 on timeout, after killing exactly the process group started here)."""
    import signal
    import socket
    import subprocess
    if not args.dry_launch:
        have = torch.cuda.device_count()          # (does not initialise the GPU)
        if have < args.gpus:
            sys.stderr.write(f"[bench] --gpus {args.gpus} but this node shows {have} GPU(s)\n")
            sys.exit(5)
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["GRAPES_BENCH_SELF_LAUNCHED"] = "1"
    sys.stderr.write("[bench] launching %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)   # the session started above: the launcher and its ranks, nothing else
        except ProcessLookupError:
            pass
        proc.wait()
        sys.stderr.write(f"[bench] the ranks did not finish within {args.launch_timeout:.0f} s: killed\n")
        sys.exit(4)
    lines = [ln for ln in out.decode(errors="replace").splitlines() if ln.strip().startswith("{")]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    rc = proc.returncode
    if rc == 0 and not lines:
        sys.stderr.write("[bench] the ranks exited 0 without a result line\n")
        rc = 6
    sys.exit(rc if rc >= 0 else 128 - rc)


def dry_launch_rank():
    """One rank of `--dry_launch`: gloo on the CPU, one all-reduce, rank 0 prints what a scaling driver would parse."""
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group(backend="gloo")
    t = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "n_ranks_seen": int(t.item()),
                          "self_launched": os.environ.get("GRAPES_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
    dist.destroy_process_group()


def spmm_algorithmic_bytes(n, e, f):
    """4·[(e+n)·F (gathered H rows incl. self-loop) + n·F (out) + (e+n) (col idx)
    + (n+1) (rowptr) + n (dinv) + F (bias)] bytes per gcn_aggregate launch."""
    return 4 * ((e + n) * f + n * f + (e + n) + (n + 1) + n + f)


class ClockProbe:
    """Per-launch durations of the two kernels the north-star names — the gather-SpMM and the XW transform — measured INSIDE
    the replayed hipGraph: while the library's kernel clock table is enabled (include/grapes_hip.h: grapes_kernel_clock_*)
    every launch of those kernels reserves a (begin, end) pair of 100 MHz s_memrealtime stamps per wavefront at capture time
    and writes them at every replay; a launch's duration is (latest end - earliest begin) over its wavefronts.  The probe
    captures a second copy of the step with the table enabled (the timed copy runs with no stamp), replays it on fresh
    batches and reads the table after each replay.  `rocprofv3 --kernel-trace` of the same command sees the same graph
    nodes; its per-dispatch duration additionally contains the dispatch ramp before the first wavefront starts."""

    def __init__(self, dev, words=1 << 22):
        from grapes_amd import _lib
        self.lib = _lib.load()
        self.table = torch.zeros(words, dtype=torch.int64, device=dev)
        self.meta = {}            # clock entry index -> (kind, meta)
        self.rate_hz = float(self.lib.grapes_kernel_clock_rate_khz()) * 1e3 or 1e8

    def enable(self):
        self.lib.grapes_kernel_clock_enable(self.table.data_ptr(), self.table.numel())

    def disable(self):
        self.lib.grapes_kernel_clock_enable(None, 0)

    def install(self):
        from grapes_amd import ops
        probe = self

        def wrap(name, kind, meta_fn):
            orig = getattr(ops, name)

            def f(*a, **k):
                i0 = probe.lib.grapes_kernel_clock_launches()
                r = orig(*a, **k)
                for i in range(i0, probe.lib.grapes_kernel_clock_launches()):
                    probe.meta[i] = (kind, meta_fn(*a, **k))
                return r
            setattr(ops, name, f)

        wrap("gcn_aggregate_gather", "spmm", lambda X, ids, prep, ind_code=None, epoch=0, num_ind=0, d_epoch=None, out=None, F=None:
             (prep, (F if F is not None else X.shape[1]) + num_ind))
        wrap("gcn_aggregate_fwd", "spmm", lambda h, prep, bias=None, relu=False, out=None: (prep, h.shape[1]))
        # ... the same aggregation with the 1-wide head's product (and the gate bits) taken from the rows (transform-first layers)
        wrap("gcn_aggregate_fwd_head", "spmm", lambda h, prep, bias, relu, head_w, want_bits=False: (prep, h.shape[1]))
        wrap("linear_bias_act_fwd", "gemm", lambda x, w, bias=None, relu=False, d_n=None, out=None: (d_n, x.shape[0], x.shape[1], w.shape[0]))
        wrap("linear_bias_act_head_fwd", "gemm", lambda x, w, bias, relu, head_w, d_n=None: (d_n, x.shape[0], x.shape[1], w.shape[0]))
        wrap("linear_bias_act_head_fwd_strided", "gemm", lambda x, w, bias, relu, head_w, d_n=None: (d_n, x.shape[0], x.shape[1], w.shape[0]))
        # the gate-bit form the sampler / log-Z nets run by default (step_graph._first_fwd): same kernel, no activation tile
        wrap("linear_relu_head_fwd_bits", "gemm", lambda x, w, bias, head_w, d_n=None: (d_n, x.shape[0], x.shape[1], w.shape[0], "bits"))
        # ... and both nets' first layers of hop 0 as ONE launch (two problems over the same rows)
        wrap("linear_relu_head_fwd_bits_pair", "gemm", lambda x, w, bias, head_w, x_b, w_b, bias_b, head_w_b, d_n=None:
             (d_n, x.shape[0], x.shape[1], w.shape[0], "bits", x_b.shape[1]))

    def entries(self):
        import ctypes as C
        out = []
        for i in range(self.lib.grapes_kernel_clock_launches()):
            name = C.create_string_buffer(64); off = C.c_int64(); pairs = C.c_int32()
            self.lib.grapes_kernel_clock_entry(i, name, C.byref(off), C.byref(pairs))
            out.append((name.value.decode(), off.value, pairs.value))
        return out

    def sizes(self, entries):
        """live (n, e) of every bound entry for the replay that just ran (device reads; outside the timed region)"""
        out = []
        for i in range(len(entries)):
            kind, meta = self.meta.get(i, (None, None))
            if kind == "spmm":
                prep = meta[0]
                n = int(prep.d_n.item()) if prep.d_n is not None else prep.n
                out.append((n, int(prep.rowptr_t[n].item())))
            elif kind == "gemm":
                out.append((int(meta[0].item()) if meta[0] is not None else meta[1],))
            else:
                out.append(None)
        return out

    def read(self, entries):
        """one replay's duration_us per entry — call after torch.cuda.synchronize()"""
        t = self.table.cpu().numpy().view(np.uint64)
        res = []
        for name, off, pairs in entries:
            st = t[off:off + 2 * pairs].reshape(-1, 2)
            live = (st[:, 1] >= st[:, 0]) & (st[:, 0] > 0)
            if not live.any():
                res.append(float("nan")); continue
            res.append(float(int(st[live, 1].max()) - int(st[live, 0].min())) / self.rate_hz * 1e6)
        return res


def load_static(path=None):
    """profiles/bench_static.json: the two per-launch quantities of the roofline kernel that need their own rocprofv3 runs
    of THIS command (profiles/refresh_r03.sh writes it): the dispatch ramp a kernel-trace duration contains on top of the
    in-kernel stamps, and the HBM traffic from the FETCH_SIZE / WRITE_SIZE counter passes.  Carries the commit it was taken at."""
    path = path or os.path.join(ROOT, "profiles", "bench_static.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def roofline_from_clock(probe, entries, replays, F_ref_out, static=None, pipelined=False):
    """replays: list of (durations_us, sizes) per replay, sizes[i] = (n, e) or (n,) of entry i read after that replay."""
    rnd = lambda r: {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items()}
    spmm_rows, gemm_rows = [], []
    for i, (name, off, pairs) in enumerate(entries):
        kind, meta = probe.meta.get(i, (None, None))
        if kind is None:
            continue
        ok = [r for r in replays if r[0][i] == r[0][i]]
        if not ok:
            continue
        us = [r[0][i] for r in ok]
        if kind == "spmm":
            f = meta[1]
            by = [spmm_algorithmic_bytes(r[1][i][0], r[1][i][1], f) for r in ok]
            ref = [spmm_algorithmic_bytes(r[1][i][0], r[1][i][1], F_ref_out) for r in ok]
            # bytes that must cross HBM at least once whatever the caches do: every destination row's own feature row, the
            # <= B + K distinct source rows (not one per edge), the records, the output
            n_m = float(np.mean([r[1][i][0] for r in ok])); e_m = float(np.mean([r[1][i][1] for r in ok]))
            spmm_rows.append(dict(position=len(spmm_rows), kernel=name, F=f, n=int(n_m), e=int(e_m), bytes=float(np.mean(by)),
                                  ref_bytes=float(np.mean(ref)), unique_bytes=4.0 * (2 * n_m * f + 12 * n_m + f),
                                  us=float(np.mean(us)), us_min=float(np.min(us)), us_max=float(np.max(us))))
        else:
            fi, fo = meta[2], meta[3]
            bits = len(meta) > 4 and meta[4] == "bits"
            fib = meta[5] if len(meta) > 5 else 0          # a second problem over the same rows in the same launch (K = fib)
            np2 = 2 if fib else 1
            pad16 = lambda k: (k + 15) // 16 * 16
            ns = [r[1][i][0] for r in ok]
            wr = (lambda n: n * (4 * ((fo + 31) // 32) + 4)) if bits else (lambda n: 4.0 * n * fo)     # gate words + head | tile
            gemm_rows.append(dict(position=len(gemm_rows), kernel=name,
                                  form=("gate bits + head" if bits else "activation tile") + (", two nets side by side" if fib else ""),
                                  n=int(np.mean(ns)), K=(fi if not fib else [fi, fib]), N=fo,
                                  flop=float(np.mean([2.0 * n * (fi + fib) * fo for n in ns])),
                                  flop_exec=float(np.mean([6 * 2.0 * n * (pad16(fi) + (pad16(fib) if fib else 0)) * fo for n in ns])),
                                  bytes=float(np.mean([4.0 * (n * fi + (fi + fib) * fo) + np2 * wr(n) for n in ns])), us=float(np.mean(us))))
    roof = mf = None
    ramp = float((static or {}).get("dispatch_ramp_us", 0.0) or 0.0)
    ramp_src = "profiles/bench_static.json (rocprofv3 --kernel-trace of this command minus the stamps, commit %s)" % (static or {}).get("commit")
    if not ramp:
        # (no rocprofv3 run of this exact command on file — another workload: the ramp is a property of the dispatch, not of the
        # data; the products command's value stands in and the line says so)
        ramp, ramp_src = DEFAULT_DISPATCH_RAMP_US, "default (the products command's measured value; no kernel trace of this command on file)"
    if spmm_rows:
        for r in spmm_rows:
            r["gbs"] = round(r["bytes"] / r["us"] / 1e3, 1); r["frac"] = round(r["gbs"] / HBM_PEAK_GBS, 4)
            r["gbs_rocprof_basis"] = round(r["bytes"] / (r["us"] + ramp) / 1e3, 1)
        big = max(r["bytes"] for r in spmm_rows)
        sel = [r for r in spmm_rows if r["bytes"] >= 0.5 * big]      # the frontier-sized launches (secondary figures)
        tb, tus = sum(r["bytes"] for r in sel), sum(r["us"] for r in sel)
        ach_sel = tb / tus / 1e3
        # THE HEADLINE: every launch of the dominant kernel in a step (the small hop-0 and classifier launches included), each
        # with the dispatch ramp a rocprofv3 --kernel-trace duration contains on top of the in-kernel stamps — the basis on
        # which `achieved` = average algorithmic bytes per launch / the rocprofv3 summary's average duration of that kernel
        same = [r for r in spmm_rows if r["kernel"] == sel[0]["kernel"]]
        carried = []
        if pipelined and len(same) >= 3:
            # In the TIMED step (prelude pipeline) the first of these launches — hop 0's, of the NEXT step — rides inside the
            # classifier's aggregation launch and the last — the classifier's — carries the next step's row-order kernel: both
            # run under the names of their pair kernels (gcn_aggregate_gather_pair_k, gather_head5_sort_pair_k), so a rocprofv3
            # summary of this command lists under THIS kernel's name only the launches in between.  The headline covers exactly
            # those — it must be reproducible from that summary —; the carried ones are listed in per_position (probe copy of
            # the step without the pipeline: every launch on its own).
            carried = [same[0]["position"], same[-1]["position"]]
            same = same[1:-1]
        ab, aus = sum(r["bytes"] for r in same), sum(r["us"] for r in same)
        ach = ab / (aus + ramp * len(same)) / 1e3
        roof = dict(bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                    positions_in_headline=[r["position"] for r in same], positions_carried_by_pair_launches=carried,
                    frac_basis="ALL launches of the kernel UNDER ITS OWN NAME in the timed step (rocprofv3 basis): sum of their algorithmic bytes / sum of "
                               "(in-kernel stamp duration + dispatch ramp) — reproducible as avg_algorithmic_bytes / the kernel's "
                               "average duration in a rocprofv3 --kernel-trace --stats summary of this command",
                    dispatch_ramp_us=round(ramp, 3), dispatch_ramp_source=ramp_src,
                    frac_all_positions_stamps=round(ab / aus / 1e3 / HBM_PEAK_GBS, 4),
                    frac_frontier_launches_stamps=round(ach_sel / HBM_PEAK_GBS, 4),
                    traffic=None, copy_ceiling=6290.0, frac_of_copy_ceiling=round(ach / 6290.0, 4), kernel=sel[0]["kernel"],
                    launches_per_step=len(same), launches_per_step_frontier=len(sel),
                    avg_launch_us=round(aus / len(same) + ramp, 2), avg_launch_us_stamps=round(aus / len(same), 2),
                    avg_launch_us_all=round(aus / len(same), 2),
                    avg_launch_us_frontier_stamps=round(tus / len(sel), 2), avg_algorithmic_bytes=int(ab / len(same)),
                    avg_algorithmic_bytes_frontier=int(tb / len(sel)),
                    avg_unique_bytes=int(sum(r["unique_bytes"] for r in same) / len(same)),
                    replays=len(replays),
                    timing="in-kernel s_memrealtime stamps (first wavefront begin -> last wavefront end) of the replayed hipGraph's "
                           "own launches, read after each replay (HIP events cannot bracket a graph node), + the dispatch ramp "
                           "rocprofv3 --kernel-trace sees before the first wavefront starts",
                    per_position=[rnd(r) for r in spmm_rows],
                    note="aggregate-first layers run the SpMM on F_in(+ind)-wide rows; ref_bytes = the reference-order "
                         "(transform-then-aggregate, %d-wide) launch of the same graph; unique_bytes = what must cross HBM once "
                         "(self rows + records + output; the <= B+K source rows are cache-resident)" % F_ref_out)
        tr = (static or {}).get("traffic")
        if tr:
            pp = tr.get("per_position") or []
            if pp and len(pp) == len(same):       # per launch like `achieved`: the average over all launches of a step
                roof["traffic"] = int(sum(q["read_bytes"] + q["write_bytes"] for q in pp) / len(pp))
                roof["traffic_frontier_launches"] = tr.get("hbm_bytes_per_launch")
            else:
                roof["traffic"] = tr.get("hbm_bytes_per_launch")
            roof["traffic_detail"] = tr
    if gemm_rows:
        for r in gemm_rows:
            r["tflops_fp32_equiv"] = round(r["flop"] / r["us"] / 1e6, 2); r["hbm_gbs"] = round(r["bytes"] / r["us"] / 1e3, 1)
            r["tflops_bf16_executed"] = round(r["flop_exec"] / r["us"] / 1e6, 1)
        big = max(r["flop"] for r in gemm_rows)
        sel = [r for r in gemm_rows if r["flop"] >= 0.5 * big]
        tus = sum(r["us"] for r in sel)
        split = os.environ.get("GRAPES_GEMM_SPLIT", "1") != "0" and all(r["kernel"].startswith("gemm_wsplit") for r in sel)
        ach32 = sum(r["flop"] for r in sel) / tus / 1e6
        aus = sum(r["us"] for r in gemm_rows)
        all32 = sum(r["flop"] for r in gemm_rows) / aus / 1e6
        ng = len(gemm_rows)
        all32_rp = sum(r["flop"] for r in gemm_rows) / (aus + ramp * ng) / 1e6        # all launches, rocprofv3 basis
        if split:
            ach_sel = sum(r["flop_exec"] for r in sel) / tus / 1e6
            allx = sum(r["flop_exec"] for r in gemm_rows) / aus / 1e6
            ach = sum(r["flop_exec"] for r in gemm_rows) / (aus + ramp * ng) / 1e6
            mf = dict(bound="mfma", achieved=round(ach, 1), peak=2500.0, unit="TFLOP/s", frac=round(ach / 2500.0, 4),
                      frac_basis="ALL launches of the kernel in a step, in-kernel stamps + dispatch ramp each (rocprofv3 basis); "
                                 "executed bf16 FLOP against the dense bf16 peak",
                      kernel="gemm_wsplit_f32_k (fp32 operands split exactly into 3 bf16 terms, 6 cross products on "
                             "v_mfma_f32_32x32x16_bf16, fp32 accumulate; bias+ReLU epilogue, 1-wide head projection from the output tiles)",
                      launches_per_step=ng, launches_per_step_frontier=len(sel), avg_launch_us=round(aus / ng + ramp, 2),
                      avg_launch_us_frontier_stamps=round(tus / len(sel), 2), dispatch_ramp_us=round(ramp, 3),
                      fp32_equivalent_tflops=round(all32_rp, 2), fp32_mfma_peak_tflops=157.3,
                      frac_fp32_equivalent_of_fp32_mfma_peak=round(all32_rp / 157.3, 4),
                      frac_all_positions_stamps=round(allx / 2500.0, 4), frac_frontier_launches_stamps=round(ach_sel / 2500.0, 4),
                      fp32_equivalent_tflops_all_positions_stamps=round(all32, 2),
                      fp32_equivalent_tflops_frontier_launches_stamps=round(ach32, 2),
                      hbm_gbs=round(sum(r["bytes"] for r in gemm_rows) / (aus + ramp * ng) / 1e3, 1),
                      per_position=[rnd(r) for r in gemm_rows],
                      note="executed bf16 FLOP (6 x 2*n*ceil16(K)*N) against the dense bf16 peak; fp32_equivalent = 2*n*K*N / time "
                           "against the 157.3 TFLOP/s fp32-MFMA peak")
        else:
            mf = dict(bound="mfma", achieved=round(all32_rp, 2), peak=157.3, unit="TFLOP/s", frac=round(all32_rp / 157.3, 4),
                      frac_basis="ALL launches of the kernel in a step, in-kernel stamps + dispatch ramp each (rocprofv3 basis)",
                      kernel=sel[0]["kernel"], launches_per_step=ng, avg_launch_us=round(aus / ng + ramp, 2),
                      dispatch_ramp_us=round(ramp, 3),
                      frac_all_positions_stamps=round(all32 / 157.3, 4), frac_frontier_launches_stamps=round(ach32 / 157.3, 4),
                      per_position=[rnd(r) for r in gemm_rows])
    return roof, mf


def build_models(F, H, C, hops, device):
    from grapes_amd.modules.gcn import GCN
    torch.manual_seed(0)
    gcn_c = GCN(F, [H] * (hops - 1) + [C]).to(device)             # BASELINE "3-layer GCN" = GCN(F,[H,H,C])
    gcn_gf = GCN(F + hops + 1, [H, 1]).to(device)                 # main.py:112-113
    gcn_z = GCN(F, [H, 1]).to(device)                             # main.py:114
    return gcn_c, gcn_gf, gcn_z


def cpu_baseline(rowptr, col, X, y, train_idx, cfg, steps, state):
    """The oracle's train_step (reference control flow, SciPy-style CSR ops in numpy, torch-CPU GCNConv
    op sequence) on the host cores: kind = "port"."""
    from oracle import grapes_oracle as O
    N, deg, maxdeg, F, C, B, K, hops = cfg
    H = state["H"]
    ncores = min(os.cpu_count() or 1, 16)      # a 1-GPU box's CPU share is 16 cores (256 threads only oversubscribe)
    torch.set_num_threads(ncores)
    indptr, indices = rowptr.cpu().numpy(), col.cpu().numpy()
    Xc, yc = X.cpu(), y.cpu()
    torch.manual_seed(0)
    c, gf, z = O.GCNRef(F, [H] * (hops - 1) + [C]), O.GCNRef(F + hops + 1, [H, 1]), O.GCNRef(F, [H, 1])
    c.load_state_dict({k: v.cpu() for k, v in state["c"].items()})
    gf.load_state_dict({k: v.cpu() for k, v in state["gf"].items()})
    z.load_state_dict({k: v.cpu() for k, v in state["z"].items()})
    oc = torch.optim.Adam(c.parameters(), lr=1e-3)
    og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=1e-4)
    tm = O.TensorMap(N)
    idx = train_idx.cpu().numpy()
    rng = np.random.default_rng(0)
    edges, t_total = 0, 0.0
    for s in range(steps + 1):
        tg = idx[(s * B) % max(1, len(idx) - B):][:B]
        t0 = time.perf_counter()
        tr = O.train_step(indptr, indices, Xc, yc, tg, c, gf, z, sampling_hops=hops, num_samples=K,
                          uniforms_fn=lambda h, n: rng.random(n, dtype=np.float32), loss_coef=1e4,
                          optimizer_c=oc, optimizer_gf=og, node_map=tm, random_sampling=state.get("random_sampling", False))
        dt = time.perf_counter() - t0
        if s == 0:
            continue                                   # first step warms the allocator / threads
        edges += tr["edges_aggregated"]
        t_total += dt
    return dict(value=round(edges / t_total, 1), unit="edges/s", cores=ncores, kind="port",
                sample=f"{steps} training steps of the same workload (same graph, batch size, hops) after 1 warm-up step; "
                       f"{t_total / steps * 1e3:.1f} ms/step",
                ms_per_step=round(t_total / steps * 1e3, 2))


def eager_dropin_ms(b, args, steps):
    """The loop a user of the reference keeps after INTEGRATION.md §2's three-import-line change: reference main.py:157-291
    over the drop-in modules (GCN autograd.Functions, sample_neighborhoods_from_probs, TensorMap ...) with exact-size tensors
    and one size read-back per hop — `step.GrapesTrainer`, NOT captured.  Same workload and optimisers; its own models."""
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.step import GrapesTrainer
    N, deg, maxdeg, F, C, B, K, hops = b.cfg
    c, gf, z = build_models(F, args.hidden_dim, C, hops, b.dev)
    oc = torch.optim.Adam(c.parameters(), lr=4.469e-4)
    og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=2.556e-5)
    tr = GrapesTrainer(DeviceGraph(b.rowptr, b.col, N), b.X, b.y, c, gf, z, sampling_hops=hops, num_samples=K,
                       loss_coef=15227.124, optimizer_c=oc, optimizer_gf=og, philox_seed=4321)
    for s in range(10):
        tr.step(b.batch(s))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        tr.step(b.batch(10 + s))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def reference_shaped_ms(b, args, steps):
    """INTEGRATION.md §2 measured: the reference's OWN loop shape (main.py:157-291: host-side O(N) boolean masks, data.x and
    the indicator matrix on the CPU, per-hop gathers + H2D copies, int64 CPU index tensors, .item() reads) over the drop-in
    modules — grapes_amd.reference_loop.ReferenceShapedLoop — i.e. what a user gets who changes the three import lines and
    nothing else.  Same workload, models and optimiser settings; its own models."""
    from grapes_amd.graph import DeviceGraph
    from grapes_amd.reference_loop import ReferenceShapedLoop
    N, deg, maxdeg, F, C, B, K, hops = b.cfg
    c, gf, z = build_models(F, args.hidden_dim, C, hops, b.dev)
    oc = torch.optim.Adam(c.parameters(), lr=4.469e-4)
    og = torch.optim.Adam(list(gf.parameters()) + list(z.parameters()), lr=2.556e-5)
    loop = ReferenceShapedLoop(DeviceGraph(b.rowptr, b.col, N), b.X.cpu(), b.y, c, gf, z, sampling_hops=hops, num_samples=K,
                               loss_coef=15227.124, optimizer_c=oc, optimizer_gf=og, device=b.dev)
    for s in range(3):
        loop.step(b.batch(s).cpu())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        loop.step(b.batch(3 + s).cpu())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


_REAL_STDOUT = 1


def _emit(res):
    sys.stdout.flush()
    os.write(_REAL_STDOUT, (json.dumps(res) + "\n").encode())       # the ONE line of this program's stdout


def select_modes(args, world, fits):
    """-> (part_mode, primary, secondary): which parallel modes a run measures.  `primary` lists the candidates for the line's
    `value` in the order they are tried — the FIRST that completes is the line (N > 1 default: the peer-mapped step, then the RCCL
    halo exchange should the hipIpc mapping be refused on this node); `secondary` are the measurements reported beside it in
    config, run AFTER the primary inside a deadline of their own (the replicated data-parallel step when graph + features fit
    a GPU four times over; with --halo peer the RCCL all-gather + all-to-all form of the halo exchange, unless --skip_rccl).
    N = 1: the single-GPU step (or one of the --force_* diagnostics), nothing beside it."""
    part_mode = "partition_adj" if args.partition_adjacency else "partition"
    first_part = "peer" if (args.halo == "peer" and not args.partition_adjacency) else part_mode
    if world == 1:
        return part_mode, (["peer"] if args.force_peer else ([part_mode] if args.force_partition else ["single"])), []
    if args.replicate:
        return part_mode, ["replicated"], []
    primary = [first_part] + ([part_mode] if first_part == "peer" else [])
    secondary = (["replicated"] if fits and not args.partition_only else []) + \
                ([part_mode] if (first_part == "peer" and not args.skip_rccl) else [])
    return part_mode, primary, secondary


def run_primary(candidates, run, *, world, peer_unmapped, guard, log):
    """Tries the primary candidates in order.  -> (mode or None, result or None, notes).  A candidate is abandoned for the next
    one only when it failed ON EVERY RANK ALIKE BEFORE ANY STEP RAN — the peer mapping refused (`peer_unmapped()`); any other
    failure ends the search (a rank that failed alone has left the others inside a collective: no further phase may start).
    `guard(kind, mode)` returns a context manager that bounds the phase's wall-clock time."""
    notes = {}
    for i, mode in enumerate(candidates):
        try:
            with guard("primary", mode):
                return mode, run(mode), notes
        except Exception as ex:                                  # noqa: BLE001
            if mode == "peer" and world > 1 and peer_unmapped() and i + 1 < len(candidates):
                log(f"[bench] peer mapping refused ({type(ex).__name__}: {ex}): the RCCL halo exchange is the primary measurement")
                notes["halo_peer_mapping"] = f"refused: {str(ex)[:160]}"
                continue
            log(f"[bench] primary phase '{mode}' failed ({type(ex).__name__}: {ex})")
            notes[mode] = f"failed: {type(ex).__name__}: {str(ex)[:200]}"
            if world == 1:
                raise
            return None, None, notes
    return None, None, notes


def run_secondary(modes, run, *, done, guard, log):
    """The measurements reported beside the primary, in order, each inside its own deadline; the first failure ends them (the
    other ranks may be inside its collective).  -> (results, notes)"""
    results, notes = {}, {}
    for mode in modes:
        if mode in done:
            continue
        try:
            with guard("secondary", mode):
                results[mode] = run(mode)
        except Exception as ex:                                  # noqa: BLE001
            log(f"[bench] secondary phase '{mode}' failed ({type(ex).__name__}: {ex}): not reported")
            notes[mode] = f"failed: {type(ex).__name__}: {str(ex)[:160]}"
            break
    return results, notes


class Bench:
    """One workload resident on this rank; `run(mode)` builds the trainer of that parallel mode, warms it up, times K
    steps between barriers and returns the whole-job numbers."""

    def __init__(self, args, world, rank, dev):
        from grapes_amd import synth
        self.args, self.world, self.rank, self.dev = args, world, rank, dev
        self.cfg = synth.CONFIGS[args.workload]
        N, deg, maxdeg, F, C, B, K, hops = self.cfg
        t0 = time.time()
        gen = torch.Generator(device=dev); gen.manual_seed(args.seed + 1)
        if N > (1 << 26):     # papers100M: chunked endpoint draws + the library's own CSR ingest (3.2e9 edges, 64-bit offsets)
            self.rowptr, self.col = synth.synth_graph_device_chunked(N, deg, maxdeg, seed=args.seed, device=dev)
            self.X = synth.randn_rows_(torch.empty(N, F, device=dev), generator=gen)
        else:
            self.rowptr, self.col = synth.synth_graph_device(N, deg, maxdeg, seed=args.seed, device=dev)   # same seed on every rank
            self.X = torch.randn(N, F, device=dev, generator=gen)
        self.y = torch.randint(0, C, (N,), device=dev, generator=gen)
        self.n_train = max(B * 4, int(0.08 * N))                  # products: 196,615 / 2,449,029 train nodes
        self.train_idx = torch.randperm(N, device=dev, generator=gen)[:self.n_train]
        self.graph_bytes = self.rowptr.numel() * 8 + self.col.numel() * 4 + self.X.numel() * 4
        self.setup_s = time.time() - t0
        self.e_cap = min(args.e_cap if args.e_cap > 0 else {"reddit": 1 << 19, "papers100m": 1 << 18}.get(args.workload, 1 << 17),
                         self.col.numel() + 1)

    def batch(self, s):   # unshuffled sequential chunks of train_idx (main.py:126), a different stripe per rank
        B = self.cfg[5]
        o = ((s * self.world + self.rank) * B) % max(1, self.n_train - B)
        return self.train_idx[o:o + B]

    def make(self, mode, models=None, seed=None, optim=True, capture=True, grad_sync="auto", pipeline=True):
        """mode: "single" | "replicated" (dp: per-GPU copy of graph + X, gradient all-reduce) | "peer" (X 1-D partitioned and
        read in place from the owners' HBM over xGMI, adjacency replicated, no exchange) | "partition" (X 1-D partitioned,
        adjacency replicated, halo all-to-all per hop) | "partition_adj" (adjacency partitioned too)."""
        from grapes_amd.graph import DeviceGraph
        from grapes_amd.step_graph import GraphedTrainer
        args, world, rank, dev = self.args, self.world, self.rank, self.dev
        N, deg, maxdeg, F, C, B, K, hops = self.cfg
        H = args.hidden_dim
        if mode in ("single", "replicated"):
            g, X_arg = DeviceGraph(self.rowptr, self.col, N), self.X
        elif mode == "peer":
            from grapes_amd.dist import partition_bounds
            from grapes_amd.peer import PeerFeatures
            if getattr(self, "_peer_x", None) is None:
                if world == 1:      # in-process shards: the table path of the kernel without a link
                    pb = partition_bounds(N, max(1, min(8, args.peer_shards)))
                    self._peer_x = PeerFeatures.from_shards([self.X[a:z].clone() for a, z in zip(pb, pb[1:])])
                else:               # collective; raises on EVERY rank when any mapping is refused
                    pb = partition_bounds(N, world)
                    self._peer_x = PeerFeatures.open(self.X[pb[rank]:pb[rank + 1]].clone(), pb, rank, world)
            g, X_arg = DeviceGraph(self.rowptr, self.col, N), self._peer_x
        else:
            from grapes_amd.dist import shard_full_graph
            maxd = int((self.rowptr[1:] - self.rowptr[:-1]).max().item())
            g = shard_full_graph(self.rowptr, self.col, self.X, rank, world, max_degree=maxd,
                                 replicate_adjacency=(mode == "partition"))
            X_arg = None
        gcn_c, gcn_gf, gcn_z = models if models is not None else build_models(F, H, C, hops, dev)
        opt_c = opt_gf = None
        if optim:
            opt_c = torch.optim.Adam(gcn_c.parameters(), lr=4.469e-4, capturable=True, fused=True)      # configs/gflownet/ogbn-products.txt
            opt_gf = torch.optim.Adam(list(gcn_gf.parameters()) + list(gcn_z.parameters()), lr=2.556e-5, capturable=True, fused=True)
        gs = None
        if grad_sync == "auto" and (mode != "single" or args.force_grad_sync):   # (peer included)
            from grapes_amd.dist import make_grad_sync
            gs = make_grad_sync(world)                            # one flat RCCL all-reduce per optimiser step
        tr = GraphedTrainer(g, X_arg, self.y, gcn_c, gcn_gf, gcn_z, batch_size=B, sampling_hops=hops, num_samples=K,
                            loss_coef=15227.124, optimizer_c=opt_c, optimizer_gf=opt_gf, e_cap=self.e_cap,
                            philox_seed=(1234 if seed is None else seed) + rank, capture=capture, grad_sync=gs,
                            random_sampling=args.random_sampling, pipeline=pipeline and not args.no_pipeline)
        return tr, g, (gcn_c, gcn_gf, gcn_z)

    def run(self, mode):
        args, world, rank, dev = self.args, self.world, self.rank, self.dev
        hops = self.cfg[7]
        trainer, g, models = self.make(mode)
        # the captured step feeds itself from the device-resident training ids (the chunks of batch(s)) and keeps the
        # edge totals on the device: no copy / cast / accumulation launch around a replay
        trainer.attach_loader(self.train_idx, stride=world, offset=rank)
        # W untimed warm-up steps.  The captured step needs its eager steps + the capture itself before it can be timed, so
        # a W smaller than that is raised to it (still untimed; reported as config.warmup_effective).
        warm = max(args.warmup, trainer.eager_steps + 2)
        chain = max(int(args.chain), 1)
        for s in range(trainer.eager_steps + 2):
            trainer.step_next()
        if chain > 1:                                             # the rest of the warm-up as the timed loop runs: chained
            trainer.run_steps(warm - trainer.eager_steps - 2, chain)
            trainer.prepare_chains(args.steps, chain)             # (no graph is instantiated inside the timed region)
        else:
            for s in range(warm - trainer.eager_steps - 2):
                trainer.step_next()
        torch.cuda.synchronize()                                  # (the next step's prelude may be in flight on its own stream)
        last_warm = trainer.out["agg_counts"].to(torch.int64)     # the last warm-up step's counters: folded into the totals LATER
        trainer.edge_totals.zero_()                               # (by the next prelude that uses the same buffers), inside the timed region
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if chain > 1:
            out = trainer.run_steps(args.steps, chain)            # EXACTLY args.steps steps, `chain` of them per hipGraphLaunch
        else:
            for s in range(args.steps):
                out = trainer.step_next()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:                                 # agree on the status first: a rank that raises alone leaves the others
            st_all = g.status.clone()                 # waiting in the next collective
            dist.all_reduce(st_all, op=dist.ReduceOp.MAX)
            if int(st_all.item()) and not int(g.status.item()):
                raise RuntimeError(f"another rank overflowed a capacity (status {int(st_all.item())}): raise --e_cap")
        trainer.check()                               # capacity overflow would have been flagged on the device
        # per graph build: edges x the aggregations that ran over it
        ev = (trainer.edge_totals - last_warm + out["agg_counts"].to(torch.int64)).cpu()    # + the last step's, not yet added
        wv = torch.tensor(out["agg_weights"], dtype=torch.int64)
        xv = torch.tensor(out["agg_executed"], dtype=torch.int64)
        edges = float((ev * wv).sum().item())
        sec = {}
        # SURVEY §8(d) secondary columns (this rank): the classifier-only term, exact over the timed steps, and the
        # self-loop-inclusive count (+ one unit self-loop per row of every GCNConv call; rows taken from the last step)
        sec["edges_classifier_per_step"] = round(float((ev[hops:] * wv[hops:]).sum().item()) / args.steps, 1)
        rows = sum(int(c.item()) * int(wv[h]) for h, c in enumerate(out["batch_counts"])) + int(out["n_all"].item()) * out["classifier_layers"]
        sec["edges_incl_self_loops_per_step"] = round(edges / args.steps + rows, 1)
        # aggregations that ran as launches: `value` counts every GCNConv forward the reference performs;
        # where one is obtained algebraically from another's result (the log-Z net's first layer reads the sampler net's
        # Â[X|ind] at hop 0) it is not a launch of its own — this count leaves those out
        sec["edges_executed_per_step"] = round(float((ev * xv).sum().item()) / args.steps, 1)
        # the hop gather-SpMM launches of the LAST timed step: rows, edges and SURVEY §8(d) bytes at the sampler net's input width —
        # so that counter traffic collected over a SHORT run of this command (profiles/make_bench_static.py: the --pmc passes cannot
        # afford a long one) is compared with the algorithmic bytes of the launches it was collected on, not of another training state
        try:
            fk = (cfg_f := self.cfg[3]) + (hops + 1 if not args.random_sampling else 0)
            fk = (fk + 3) // 4 * 4
            lastc = out["agg_counts"].cpu()
            sec["last_step_gather"] = [dict(hop=h, n=int(c.item()), e=int(lastc[h]), bytes=int(spmm_algorithmic_bytes(int(c.item()), int(lastc[h]), fk)))
                                       for h, c in enumerate(out["batch_counts"])]
        except Exception:
            pass
        t_el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        t_ed = torch.tensor([edges, float((ev * xv).sum().item())], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
            dist.all_reduce(t_ed, op=dist.ReduceOp.SUM)
        elapsed, edges, edges_x = float(t_el.item()), float(t_ed[0].item()), float(t_ed[1].item())
        res = dict(mode=mode, value=round(edges / elapsed, 1), value_executed=round(edges_x / elapsed, 1),
                   ms_per_step=round(elapsed / args.steps * 1e3, 3), edges_per_step_per_gpu=round(edges / args.steps / world, 1),
                   warm=warm, secondary=sec,
                   segments=(trainer.graph_obj.num_segments if trainer.graph_obj is not None else 0),
                   collectives_per_step=(trainer.graph_obj.num_collectives if trainer.graph_obj is not None else 0),
                   exchanged_mb_per_step_per_gpu=(round(getattr(trainer, "_bytes_per_step", 0) / 2**20, 2)),
                   prelude_riders=([dict(rode=st.riders[0], alone=st.riders[1]) for st in trainer._sets]
                                   if getattr(trainer, "_sets", None) else None),
                   # steps per hipGraphLaunch in the timed loop, as (steps, kernel nodes) of the chains that exist (1: none was used)
                   steps_per_graph_launch=(max([k[1] for k in trainer._chains if k[0] != "b"] + [1])),
                   chain_nodes=sorted({c.nodes for c in trainer._chains.values()}),
                   # N > 1: a step's last graph segment and the next step's first as one launch (the collectives stay host-issued)
                   boundary_chains=sum(1 for k in trainer._chains if k[0] == "b"))
        return res, trainer, g, models


def main():
    # stdout carries exactly one JSON line: everything else that writes to fd 1 (RCCL prints a version banner there under
    # NCCL_DEBUG=VERSION, libraries print warnings) is sent to stderr for the whole run.
    global _REAL_STDOUT
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                        # never returns
    if args.dry_launch:
        if "WORLD_SIZE" not in os.environ:
            os.environ.update(WORLD_SIZE="1", RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29511")
        dry_launch_rank()
        return
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    from grapes_amd import _lib
    _lib.load()
    if (args.force_partition or args.force_grad_sync or args.force_peer) and world == 1 and not dist.is_initialized():
        import socket
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(port))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)

    b = Bench(args, world, rank, dev)
    N, deg, maxdeg, F, C, B, K, hops = b.cfg
    H = args.hidden_dim
    nnz = b.col.numel()
    hbm_gib = torch.cuda.get_device_properties(dev).total_memory / 2**30
    fits = b.graph_bytes <= torch.cuda.get_device_properties(dev).total_memory // 4

    # ---- which parallel modes run (select_modes).  N = 1: the single-GPU step.  N > 1 (BASELINE config 4: "1-D node partition
    # on 8 x MI355X with xGMI halo all-to-all"): the PRIMARY line is the partitioned step and it is measured FIRST.  With --halo
    # peer (default) that is the PEER-MAPPED step: every rank maps the other ranks' feature shards (hipIpc) and the gather kernels
    # read halo rows in place over xGMI — no exchange, one collective (the gradient all-reduce), two graph segments; if the mapping
    # is refused on this node (on every rank alike, before a step ran) the RCCL all-gather + all-to-all form becomes the primary
    # and config says so.  Then rank 0's roofline probe under that mode.  Only THEN the measurements reported beside it — the
    # replicated data-parallel step (per-GPU copy of graph + features, gradient all-reduce only; skipped when the data do not fit a
    # GPU four times) and the RCCL form of the halo exchange — each inside --partition_deadline seconds: a secondary phase that
    # hangs or fails costs only itself (the primary's line is printed, exit 0).  A primary that does not finish: exit 3.
    part_mode, prim_cands, sec_modes = select_modes(args, world, fits)
    peer_note = {}
    results = {}

    def line(primary, extra_cfg=None, roof=None, roof_mfma=None, cpu=None, median_ms=None, mean_ev_ms=None, status="ok"):
        r = results[primary]
        mode_txt = {
            "single": ("single GPU; one captured hipGraph per step" + ("" if args.no_pipeline else
                       "; the NEXT step's weight-independent prelude (next batch, hop 0's expansion, compaction, graph build and "
                       "gather-SpMM) rides as extra workgroups in this step's hop-1 launches of the same kernels (two sets of "
                       "scratch tables, alternating)")),
            "replicated": (f"dp{world}: independent mini-batches per GPU over a per-GPU copy of graph + features "
                           f"({b.graph_bytes / 2**30:.1f} GiB of {hbm_gib:.0f} GiB HBM), one flat gradient all-reduce per optimiser step "
                           "(RCCL over xGMI); step captured as hipGraph segments around it"),
            "peer": (f"dp{world} mini-batches over a 1-D node partition of the feature matrix (X sharded "
                     f"{world if world > 1 else str(args.peer_shards) + ' in-process'} ways, adjacency replicated: get_neighborhoods is local); "
                     "every rank maps the other ranks' shards (hipIpc) and the fused gather-SpMM reads each halo row IN PLACE from the "
                     "HBM of the GPU that owns it (xGMI loads, shard picked per row from a register table) — no request/reply exchange, "
                     "no data-path collective; one flat gradient all-reduce per optimiser step (RCCL); the step is one hipGraph segment "
                     "up to the all-reduce and one after it"),
            "partition": (f"dp{world} mini-batches over a 1-D node partition of the feature matrix (X sharded {world} ways, adjacency "
                          "replicated: get_neighborhoods is local); per hop: all-gather of the id lists + ONE all-to-all of halo "
                          "feature rows in fixed slots (the classifier's rows are found among them: no request of their own); one flat "
                          "gradient all-reduce per optimiser step (RCCL over xGMI); step captured as hipGraph segments between the collectives"),
            "partition_adj": (f"dp{world} mini-batches over a 1-D node partition of CSR + X ({world} ways); per hop: all-gather of query lists "
                              "+ all-to-all of adjacency rows and of halo feature rows in fixed slots; one flat gradient all-reduce per "
                              "optimiser step (RCCL over xGMI); hipGraph segments between the collectives"),
        }[primary]
        cfg = {"workload": f"{args.workload}-like synthetic graph N={N} nnz={nnz} F={F} C={C}; "
                           f"B={B} targets/step/GPU, {hops} sampling hops x K={K} nodes, H={H}; "
                           f"sampler GCN(F+{hops + 1},[{H},1]), log-Z GCN(F,[{H},1]), classifier GCN(F,[{H}]*{hops - 1}+[{C}]); TB loss, Adam x2",
               "parallelism": mode_txt, "edges_per_step_per_gpu": r["edges_per_step_per_gpu"], **r["secondary"],
               "value_executed_edges_per_s": r["value_executed"], "setup_s": round(b.setup_s, 1), "warmup_effective": r["warm"],
               "graph_segments_per_step": r["segments"], "steps_per_graph_launch": r.get("steps_per_graph_launch", 1), "chain_nodes": r.get("chain_nodes"), "boundary_chains": r.get("boundary_chains", 0), "collectives_per_step": r["collectives_per_step"],
               "prelude_launches_per_step": r.get("prelude_riders"),
               "exchanged_MiB_per_step_per_gpu": r["exchanged_mb_per_step_per_gpu"],
               "n_ranks_seen": (dist.get_world_size() if dist.is_initialized() else 1)}
        for k, v in results.items():
            if k != primary:
                cfg[{"replicated": "replicated_dp", "partition": "partition", "partition_adj": "partition_adj", "single": "single",
                     "peer": "peer"}[k]] = \
                    dict(value=v["value"], ms_per_step=v["ms_per_step"], collectives_per_step=v["collectives_per_step"])
        cfg.update(peer_note)
        cfg.update(extra_cfg or {})
        if args.random_sampling:
            cfg["workload"] += "  [--random_sampling: uniform exact-k draws, classifier only (reference configs/random/*)]"
        return {
            "metric": "sampled edges aggregated/sec, ogbn-products 3-layer GFlowNet" if (args.workload == "products" and not args.random_sampling) else
                      f"sampled edges aggregated/sec, {args.workload}-shaped {hops}-layer GFlowNet (not the BASELINE headline workload)",
            "value": r["value"], "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": r["ms_per_step"],
            "ms_per_step_median": None if median_ms is None else round(median_ms, 4),
            "ms_per_step_mean_event_timed": None if mean_ev_ms is None else round(mean_ev_ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": cfg, "roofline": roof, "roofline_mfma": roof_mfma, "cpu_baseline": cpu, "status": status,
        }

    log = lambda m: sys.stderr.write(m + "\n")
    state = dict(primary=None, roof=None, roof_mfma=None, median=None, mean_ev=None, extra=None)

    class Guard:
        """Bounds a phase's wall-clock time on N > 1 (an untested fabric, a hung collective — every rank runs the same guard
        with the same deadline, so all of them leave).  A PRIMARY phase that does not finish: a failure line (no value), exit 3.
        A SECONDARY phase that does not finish costs only itself: rank 0 prints the primary's line — measured before any
        secondary started, roofline included — with a note, and every rank exits 0.  Nothing is ever re-exec'ed."""

        def __init__(self, kind, mode):
            self.kind, self.mode, self.ev, self.th = kind, mode, None, None

        def __enter__(self):
            if world > 1 and args.partition_deadline > 0:
                import threading
                self.ev = threading.Event()
                self.th = threading.Thread(target=self._wait, daemon=True)
                self.th.start()
            return self

        def __exit__(self, *exc):
            if self.ev is not None:
                self.ev.set()
            return False

        def _wait(self):
            if self.ev.wait(args.partition_deadline):
                return
            log(f"[bench] {self.kind} phase '{self.mode}' exceeded {args.partition_deadline}s")
            if self.kind in ("secondary", "roofline") and state["primary"] is not None:
                key = "roofline_probe" if self.kind == "roofline" else {"replicated": "replicated_dp"}.get(self.mode, self.mode)
                peer_note[key] = "did not finish within the deadline (not measured)"
                if rank == 0:
                    _emit(line(state["primary"], extra_cfg=state["extra"], roof=state["roof"], roof_mfma=state["roof_mfma"]))
                os._exit(0)
            if rank == 0:
                fb = next((m for m in results if m != self.mode), None)
                if fb is not None:            # (a fallback measurement exists: print it for a human; the exit code says failed)
                    _emit(line(fb, {self.mode: "did not finish within the deadline"}, status="partition_failed"))
                else:
                    _emit({"metric": "sampled edges aggregated/sec, ogbn-products 3-layer GFlowNet", "value": None, "unit": "edges/s",
                           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "status": f"{self.kind}_timeout",
                           "config": {"workload": args.workload, "phase": self.mode, **peer_note}})
            os._exit(3)

    # ---- the PRIMARY measurement first: nothing that runs later can cost it
    kept = {}

    def run(mode):
        res, tr_, g_, models_ = b.run(mode)
        kept[mode] = (tr_, g_, models_)
        return res
    primary, res, pn = run_primary(prim_cands, run, world=world, peer_unmapped=lambda: getattr(b, "_peer_x", None) is None,
                                   guard=Guard, log=log)
    peer_note.update(pn)
    if primary is None:
        # every candidate failed (on all ranks alike, or the guards above end the run): the replicated step, if it can be
        # measured, is printed for a human and the exit code says that the partitioned step was not measured
        if "replicated" in sec_modes:
            r2, n2 = run_secondary(["replicated"], run, done=results, guard=Guard, log=log)
            results.update(r2)
            if "replicated" in results and rank == 0:
                _emit(line("replicated", {"partition": "; ".join(f"{k}: {v}" for k, v in peer_note.items())}, status="partition_failed"))
        os._exit(3)
    results[primary] = res
    state["primary"] = primary
    trainer, g, models = kept[primary]
    state["extra"] = {"hbm": dict(graph_and_features_GiB=round(b.graph_bytes / 2**30, 2),
                                  peak_allocated_GiB=round(torch.cuda.max_memory_allocated(dev) / 2**30, 2),
                                  device_total_GiB=round(hbm_gib, 1))}

    # ---- per-step times: a separate, event-timed pass over the same replays (an event record between two graph launches;
    # not inside the timed region above, whose value stays free of them).  SURVEY §8(d): median of >= 100 steps.
    median_ms = mean_ev_ms = None
    if world == 1 and not args.no_median:
        nm = max(100, min(args.steps, 500))
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(nm + 1)]
        evs[0].record()
        for s in range(nm):
            trainer.step_next()
            evs[s + 1].record()
        torch.cuda.synchronize()
        trainer.check()
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(nm))
        median_ms, mean_ev_ms = per[nm // 2], sum(per) / nm

    # ---- roofline of the two kernels the north-star names, under the PRIMARY mode (N > 1: every rank replays its probe copy
    # of the step — with peer-mapped shards that is the gather under remote loads from all ranks at once; rank 0 reports)
    roof = roof_mfma = None
    if (not args.no_roofline) and not args.random_sampling:
        # A second copy of the captured step with the kernel clock table enabled: its launches of the gather-SpMM and of the
        # XW GEMM stamp begin / end per wavefront at every replay (ClockProbe above).  Same graph, shapes, weights, data.
        with Guard("roofline", primary):
            probe = ClockProbe(dev)
            probe.install()
            nprobe = 20
            # (one hipGraph per step here: the probe reads per-launch stamps after every replay, nothing is pipelined)
            ptr, _, _ = b.make(primary, models=models, seed=99, optim=False, grad_sync=None, pipeline=False)
            ptr.attach_loader(b.train_idx, stride=world, offset=rank if world > 1 else 7)
            for s in range(ptr.eager_steps):
                ptr.step_next()
            torch.cuda.synchronize()
            probe.enable()
            try:
                ptr.step_next()                                   # capture (reserves the stamp ranges) + first replay
                torch.cuda.synchronize()
                entries = probe.entries()
                replays = []
                for s in range(nprobe):
                    probe.table.zero_()                           # (a workgroup beyond the live row count leaves no stamp)
                    ptr.step_next()
                    torch.cuda.synchronize()
                    replays.append((probe.read(entries), probe.sizes(entries)))
                ptr.check()
                # (the static file holds the counter passes / dispatch ramp of the single-GPU products command only)
                roof, roof_mfma = roofline_from_clock(probe, entries, replays, H,
                                                      static=load_static() if (args.workload == "products" and primary == "single") else None,
                                                      pipelined=getattr(trainer, "_sets", None) is not None)
            finally:
                probe.disable()
            if roof is not None:
                roof["mode"] = primary
                if world > 1:
                    roof["note_multi_gpu"] = (f"rank 0's launches while all {world} ranks replay their steps"
                                              + ("; halo rows are loads from the owners' HBM over xGMI" if primary == "peer" else ""))
            if roof is not None and median_ms:
                tot_b = sum(r["bytes"] for r in roof["per_position"])
                roof["step_level"] = dict(spmm_algorithmic_bytes_per_step=int(tot_b), ms_per_step_median=round(median_ms, 4),
                                          spmm_bytes_over_step_time_gbs=round(tot_b / (median_ms * 1e-3) / 1e9, 1),
                                          note="algorithmic bytes of the forward gather-SpMM launches only, over the WHOLE step time")
            del ptr
    state.update(roof=roof, roof_mfma=roof_mfma)

    # ---- the measurements reported beside the primary, each inside its own deadline
    if sec_modes:
        if world > 1:
            dist.barrier()
        r2, n2 = run_secondary(sec_modes, run, done=results, guard=Guard, log=log)
        results.update(r2)
        peer_note.update({{"replicated": "replicated_dp"}.get(k, k): v for k, v in n2.items()})

    cpu = None
    if rank == 0 and world == 1 and args.cpu_steps > 0 and primary == "single":
        cst = dict(H=H, c=None, gf=None, z=None, random_sampling=args.random_sampling)
        torch.manual_seed(0)
        c0, gf0, z0 = build_models(F, H, C, hops, "cpu")          # the step's initial weights (same seed)
        cst.update(c=c0.state_dict(), gf=gf0.state_dict(), z=z0.state_dict())
        cpu = cpu_baseline(b.rowptr, b.col, b.X, b.y, b.train_idx, b.cfg, args.steps if args.steps < args.cpu_steps else args.cpu_steps, cst)

    extra = state["extra"]
    if rank == 0 and world == 1 and primary == "single" and args.eager_steps > 0 and not args.random_sampling:
        try:
            ems = eager_dropin_ms(b, args, args.eager_steps)
            extra["eager_dropin_loop"] = dict(
                ms_per_step=round(ems, 3), steps=args.eager_steps, ratio_to_captured=round(ems / results[primary]["ms_per_step"], 2),
                what="reference main.py loop over the drop-in modules (step.GrapesTrainer: autograd.Functions, exact-size tensors, "
                     "one size read-back per hop; INTEGRATION.md section 2) — the cost of staying drop-in; `value` is the captured step")
        except Exception as ex:                                   # noqa: BLE001  (a side measurement must not lose the line)
            sys.stderr.write(f"[bench] eager drop-in loop failed: {type(ex).__name__}: {ex}\n")
            extra["eager_dropin_loop"] = dict(error=f"{type(ex).__name__}: {str(ex)[:160]}")
        try:
            nrs = max(3, min(args.eager_steps, 10))
            rms = reference_shaped_ms(b, args, nrs)
            extra["reference_shaped_dropin_loop"] = dict(
                ms_per_step=round(rms, 2), steps=nrs, ratio_to_captured=round(rms / results[primary]["ms_per_step"], 1),
                what="reference main.py:157-291 with ONLY the three import lines of INTEGRATION.md section 2 changed "
                     "(grapes_amd.reference_loop): O(N) boolean masks, data.x and the indicator matrix on the host, per-hop "
                     "gathers + H2D copies, .item() reads — the drop-in modules underneath; what a user of the reference "
                     "gets before moving the data to the device (then: eager_dropin_loop) or the step into the graph (value)")
        except Exception as ex:                                   # noqa: BLE001
            sys.stderr.write(f"[bench] reference-shaped loop failed: {type(ex).__name__}: {ex}\n")
            extra["reference_shaped_dropin_loop"] = dict(error=f"{type(ex).__name__}: {str(ex)[:160]}")
    if rank == 0:
        _emit(line(primary, extra_cfg=extra, roof=roof, roof_mfma=roof_mfma, cpu=cpu, median_ms=median_ms, mean_ev_ms=mean_ev_ms))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
