#!/usr/bin/env python3
"""What bounds the gather-SpMM at the step's hop shape?  The same launch (in-kernel clock stamps, first wavefront begin ->
last wavefront end) with (A) the step's random feature rows read cold from HBM, (B) the same rows warm in the memory-side
cache (second of two back-to-back launches), (C) cold but CONSECUTIVE feature rows (ids = a contiguous range)."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from grapes_amd import ops, synth, _lib
from grapes_amd.graph import DeviceGraph
lib = _lib.load_diag()
dev = torch.device("cuda", 0)
N, deg, maxdeg, F, C_, B, K, hops = synth.CONFIGS["products"]
rowptr, col = synth.synth_graph_device(N, deg, maxdeg, seed=0, device=dev)
g = DeviceGraph(rowptr, col, N)
gen = torch.Generator(device=dev); gen.manual_seed(1)
X = torch.randn(N, F, device=dev, generator=gen)
Xp = ops.pad_features(X)[0]
prev = torch.randperm(N, device=dev, generator=gen)[:768].to(torch.int32)
e_cap = 1 << 17; n_cap = e_cap + 513
eoff, d_e = ops.frontier_offsets(g.rowptr, prev)
src, dst, _ = ops.frontier_expand(g.rowptr, g.col, prev, eoff, e_cap, status=g.status)
ops.bitmap_mark(g.prev_bits, None, prev, N, status=g.status)
ops.bitmap_mark_rows(g.bits, g.bits1, prev, eoff, N, status=g.status)
ops.bitmap_mark(g.bits, g.bits1, dst, N, d_n=d_e, status=g.status)
batch, neigh, nbl, counts = ops.frontier_compact(g.bits, g.bits1, g.prev_bits, N, n_cap, node_map=g.node_map, status=g.status)
d_nb = counts[0:1]
ep = torch.ones(1, dtype=torch.int32, device=dev)
ops.indicator_mark(g.ind_code, neigh, 0, 2, d_n=counts[1:2], d_epoch=ep)
seq = (torch.arange(n_cap, dtype=torch.int32, device=dev) + 100000).contiguous()
heads = ops.PreparedGraph(src, dst, n_cap, d_n=d_nb, d_e=d_e, status=g.status, src_grouped=True, items_fwd=False, node_map=g.node_map, head_ids=batch)
heads_seq = ops.PreparedGraph(src, dst, n_cap, d_n=d_nb, d_e=d_e, status=g.status, src_grouped=True, items_fwd=False, node_map=g.node_map, head_ids=seq)
n, e = int(d_nb), int(d_e)
print(f"frontier rows {n}, edges {e}")
table = torch.zeros(1 << 22, dtype=torch.int64, device=dev)
rate = float(lib.grapes_kernel_clock_rate_khz()) * 1e3
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)
out = torch.empty((n_cap, 104), device=dev)
def stamp_us():
    torch.cuda.synchronize()
    i = lib.grapes_kernel_clock_launches() - 1
    name = C.create_string_buffer(64); off = C.c_int64(); pairs = C.c_int32()
    lib.grapes_kernel_clock_entry(i, name, C.byref(off), C.byref(pairs))
    t = table[off.value: off.value + 2 * pairs.value].cpu().numpy().reshape(-1, 2)
    t = t[(t[:, 0] > 0) & (t[:, 1] > 0)]
    return (t[:, 1].max() - t[:, 0].min()) / rate * 1e6
def launch(ids, prep):
    table.zero_()
    lib.grapes_kernel_clock_enable(table.data_ptr(), table.numel())
    ops.gcn_aggregate_gather(Xp, ids, prep, g.ind_code, 0, 4, d_epoch=ep, out=out, F=F)
    us = stamp_us()
    lib.grapes_kernel_clock_enable(None, 0)
    return us
alg = 4 * ((e + n) * 104 + n * 104) + 48 * n
res = {"A cold random": [], "B warm random": [], "C cold consecutive": [], "D warm consecutive": []}
for r in range(12):
    flush.zero_(); res["A cold random"].append(launch(batch, heads))
    res["B warm random"].append(launch(batch, heads))
    flush.zero_(); res["C cold consecutive"].append(launch(seq, heads_seq))
    res["D warm consecutive"].append(launch(seq, heads_seq))
for k, v in res.items():
    m = float(np.median(v[2:]))
    print(f"{k:20s} {m:7.2f} us   {alg / m / 1e3:7.1f} GB/s algorithmic")

if hasattr(lib, "grapes_stamp_set_spmm"):      # diagnostic build (GRAPES_LIB_PATH=.../libgrapes_hip_stamps.so): phases of the first workgroups
    buf = torch.zeros(64 * 16, dtype=torch.int64, device=dev)
    lib.grapes_stamp_set_spmm.argtypes = [C.c_void_p]
    lib.grapes_stamp_set_spmm(buf.data_ptr())
    for tag, cold in (("warm", False), ("cold", True)):
        acc = []
        for r in range(10):
            if cold: flush.zero_()
            else: ops.gcn_aggregate_gather(Xp, batch, heads, g.ind_code, 0, 4, d_epoch=ep, out=out, F=F)
            buf.zero_(); ops.gcn_aggregate_gather(Xp, batch, heads, g.ind_code, 0, 4, d_epoch=ep, out=out, F=F); torch.cuda.synchronize()
            acc.append(buf.cpu().numpy().reshape(64, 16).astype(np.float64))
        st = np.stack(acc)
        print(tag, "phases since the workgroup's own start (us, median over the first 64 workgroups x 10 launches):")
        for sl in list(range(1, 13)) + [14]:
            ok = (st[:, :, sl] > 0) & (st[:, :, 0] > 0)
            if ok.any():
                d = ((st[:, :, sl] - st[:, :, 0]) / rate * 1e6)[ok]
                it, ph = divmod(sl - 1, 3)
                what = "end" if sl == 14 else f"iteration {it} " + ("top (barrier passed)", "rows done (loads + stores drained)", "hub barrier passed")[ph]
                print(f"   slot {sl:2d} {what:48s} {np.median(d):7.2f}")

# ---- reference points at the same size: what do trivial kernels reach on ~50 MB?  (event-timed, back-to-back launches: warm)
def ev(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps * 1e3
src_rows = torch.randn(2 * n, 104, device=dev); dst_rows = torch.empty(n, 104, device=dev)
t = ev(lambda: dst_rows.copy_(src_rows[:n]))
print(f"torch copy of {n} x 104 floats (read {n*416/1e6:.1f} MB + write same): {t:.2f} us -> {2*n*416/t/1e3:.0f} GB/s")
t = ev(lambda: torch.add(src_rows[:n], src_rows[n:], out=dst_rows))
print(f"torch add of two [{n},104] (read 2x, write 1x = {3*n*416/1e6:.1f} MB): {t:.2f} us -> {3*n*416/t/1e3:.0f} GB/s")
t = ev(lambda: ops.gcn_aggregate_gather(Xp, batch, heads, g.ind_code, 0, 4, d_epoch=ep, out=out, F=F))
print(f"gather-SpMM, event-timed back-to-back (warm): {t:.2f} us -> {alg/t/1e3:.0f} GB/s algorithmic")
idx = batch[:n].long()
t = ev(lambda: torch.index_select(X, 0, idx, out=dst_rows[:, :100].contiguous()) if False else torch.index_select(X, 0, idx))
print(f"torch index_select of the same {n} rows of X (read + write {2*n*400/1e6:.1f} MB): {t:.2f} us -> {2*n*400/t/1e3:.0f} GB/s")

# ---- the ingredients priced one at a time (measurement-only kernels in the library: grapes_debug_gather_probe)
lib.grapes_debug_gather_probe.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
def probe(variant, cap, cold):
    if cold: flush.zero_()
    table.zero_()
    lib.grapes_kernel_clock_enable(table.data_ptr(), table.numel())
    rc = lib.grapes_debug_gather_probe(Xp.data_ptr(), Xp.shape[1], heads.row_head.data_ptr(), out.data_ptr(), n, 104, variant, cap, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    us = stamp_us()
    lib.grapes_kernel_clock_enable(None, 0)
    return us
for variant, cap, what in ((0, 0, "32 lanes/row, 2 row loads, one batch per workgroup"), (1, 0, "32 lanes/row, 5 row loads, one batch per workgroup"),
                           (2, 1536, "32 lanes/row, 2 loads, 1536 resident workgroups"), (3, 1536, "32 lanes/row, 5 loads, 1536 resident workgroups"),
                           (4, 0, "64 lanes/row, 2 loads, one batch per workgroup"), (6, 2048, "64 lanes/row, 2 loads, 2048 resident workgroups")):
    w = [probe(variant, cap, False) for _ in range(8)]; c = [probe(variant, cap, True) for _ in range(8)]
    print(f"probe {variant}: {what:52s} warm {np.median(w[2:]):6.2f} us   cold {np.median(c[2:]):6.2f} us")
