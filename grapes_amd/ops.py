"""Tensor-level wrappers over the C-ABI (include/grapes_hip.h).

PyTorch is used here only as plumbing: device memory (caching allocator), the current HIP stream
and dtype/shape checks.  Every function enqueues HIP kernels on torch's current stream and
returns without synchronising; `d_*` arguments are optional int32 device scalars carrying the
true sizes of capacity-padded buffers (see the header for the convention).
"""
from __future__ import annotations

from typing import List, Optional

import os

import torch

from . import _lib
from ._lib import diag_switch as _sw      # A/B switches: the default unless GRAPES_DIAG=1 (the product has one configuration)

_i32 = torch.int32
_i64 = torch.int64
_f32 = torch.float32


def lib():
    return _lib.load()


def _stream() -> int:
    """torch's current HIP stream as a raw pointer.  (torch.cuda.current_stream() builds a Stream object through several
    Python layers: ~10 us per call, 70 calls per eager step — profiles/eager_profile.py; the raw query is one C call.)"""
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:          # (another torch build: the public, slower way)
        return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk(t: Optional[torch.Tensor], dtype, name: str, allow_none: bool = False):
    if t is None:
        if allow_none:
            return
        raise TypeError(f"{name} is required")
    if not t.is_cuda:
        raise _lib.GrapesHipError(f"{name} must live in HBM (cuda tensor); grapes_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


_KEEP = [None]        # while launches are RECORDED as riders (rider_keep): the temporaries they will use later must outlive the call


def rider_keep(on: bool):
    """on=True: from now on every scratch tensor of the wrappers below is also appended to a list (launches recorded by
    grapes_rider_record_begin run LATER, every step: a workspace freed when the wrapper returns would be somebody else's memory
    by then); on=False: stop and return that list — the caller keeps it for as long as the recorded program lives."""
    if on:
        _KEEP[0] = []
        return None
    kept, _KEEP[0] = _KEEP[0], None
    return kept


def _tmp(t: torch.Tensor) -> torch.Tensor:
    if _KEEP[0] is not None:
        _KEEP[0].append(t)
    return t


def _ws(nbytes: int, device) -> torch.Tensor:
    return _tmp(torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device))


_SYNC = {}
_ONE_LAUNCH_PREP = _sw("GRAPES_ONE_LAUNCH_PREP", "1") != "0"      # A/B switches for the look-back forms
_ONE_LAUNCH_SLICE = _sw("GRAPES_ONE_LAUNCH_SLICE", "0") != "0"    # measured slower (4 edges per thread): off
_PREFETCH_ROWS = _sw("GRAPES_PREFETCH_ROWS", "1") != "0"          # A/B: the hop build touches the gather-SpMM's rows of X
# GRAPES_PREP_FUSED=1 (read by the library): the grouped, pre-zeroed hop-graph build as ONE cooperative launch with grid
# barriers instead of four launches — measured slower (profiles/r03_prep_fused_ab.txt): off


_LANE = [0]


def set_scratch_lane(lane: int) -> int:
    """Selects which copy of the per-device `sync` scratch the next calls bake into their launches; returns the previous lane.
    Launches that may run AT THE SAME TIME on two streams (step_graph's prelude pipeline: the next step's weight-independent
    index chain under the current step) must not share it; within a lane calls are stream-ordered."""
    old, _LANE[0] = _LANE[0], int(lane)
    return old


def sync_scratch(device) -> torch.Tensor:
    """The per-device `sync` scratch of the one-launch scans (include/grapes_hip.h: GRAPES_SYNC_WORDS): zero at rest,
    left zero by every kernel that uses it.  Shared by all calls of a lane on the device (set_scratch_lane), which therefore
    have to be stream-ordered (they are: the index pipeline of a lane runs on one stream); pass your own `sync=` tensor
    otherwise."""
    dev = torch.device(device)
    t = _SYNC.get((dev, _LANE[0]))
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("sync_scratch: first use inside a stream capture; run one eager step first")
        t = torch.zeros(256, dtype=_i64, device=dev)
        _SYNC[(dev, _LANE[0])] = t
    return t


# ------------------------------------------------------------------------------- learned node embeddings (--embed_nodes)
def scatter_rows(dst, ids, src, d_n=None, accumulate=False, atomic=False, F=None):
    """dst[ids[i], :F] (+)= src[i, :F]  (backward of a feature-row gather into a dense gradient; main.py:89-100,256)."""
    _chk(dst, _f32, "dst"); _chk(ids, _i32, "ids"); _chk(d_n, _i32, "d_n", True)
    if src.dtype != _f32 or not src.is_cuda or src.stride(1) != 1:
        raise TypeError("scatter_rows: src must be a float32 cuda matrix with unit column stride")
    F = int(min(dst.shape[1], src.shape[1]) if F is None else F)
    _lib.check(lib().grapes_scatter_rows(_p(dst), dst.stride(0), _p(ids), _p(src), src.stride(0), F, ids.numel(), _p(d_n),
                                         1 if accumulate else 0, 1 if atomic else 0, _stream()), "scatter_rows")
    return dst


class GatherRowsFn(torch.autograd.Function):
    """x = [X[ids] | indicators]  with a gradient for X (learned node embeddings, main.py:89-100): forward = gather_rows,
    backward = the rows' gradients added into a dense [N, F] gradient (float atomics: an id may repeat)."""

    @staticmethod
    def forward(ctx, X, ids, ind_code, epoch, num_ind):
        ctx.save_for_backward(ids)
        ctx.shape = X.shape
        return gather_rows(X.detach(), ids, ind_code, epoch, num_ind)

    @staticmethod
    def backward(ctx, dx):
        (ids,) = ctx.saved_tensors
        g = torch.zeros(ctx.shape, dtype=_f32, device=dx.device)
        if ids.numel():
            scatter_rows(g, ids, dx.contiguous(), atomic=True, F=ctx.shape[1])
        return g, None, None, None, None


# ------------------------------------------------------------------------------- TensorMap
def tensormap_update(map_t, keys, d_n=None):
    _chk(map_t, _i32, "map"); _chk(keys, _i32, "keys"); _chk(d_n, _i32, "d_n", True)
    _lib.check(lib().grapes_tensormap_update(_p(map_t), _p(keys), keys.numel(), _p(d_n), _stream()), "tensormap_update")


def tensormap_map(map_t, keys, out=None, d_n=None):
    _chk(map_t, _i32, "map"); _chk(keys, _i32, "keys"); _chk(d_n, _i32, "d_n", True)
    if out is None:
        out = torch.empty_like(keys)
    _chk(out, _i32, "out")
    _lib.check(lib().grapes_tensormap_map(_p(map_t), _p(keys), _p(out), keys.numel(), _p(d_n), _stream()), "tensormap_map")
    return out


# ------------------------------------------------------------------------------- frontier
def frontier_offsets(rowptr, nodes, d_m=None):
    """eoff int32[m+1] (eoff[m] = e) and a 1-element device tensor holding e."""
    _chk(rowptr, _i64, "rowptr"); _chk(nodes, _i32, "nodes"); _chk(d_m, _i32, "d_m", True)
    m = nodes.numel()
    eoff = torch.empty(m + 1, dtype=_i32, device=nodes.device)
    d_e = torch.empty(1, dtype=_i32, device=nodes.device)
    _lib.check(lib().grapes_frontier_offsets(_p(rowptr), _p(nodes), m, _p(d_m), _p(eoff), _p(d_e), _stream()),
               "frontier_offsets")
    return eoff, d_e


def frontier_expand(rowptr, col, nodes, eoff, e_cap, d_m=None, want_pos=False, status=None):
    _chk(rowptr, _i64, "rowptr"); _chk(col, _i32, "col"); _chk(nodes, _i32, "nodes"); _chk(eoff, _i32, "eoff")
    dev = nodes.device
    src = torch.empty(e_cap, dtype=_i32, device=dev)
    dst = torch.empty(e_cap, dtype=_i32, device=dev)
    pos = torch.empty(e_cap, dtype=_i32, device=dev) if want_pos else None
    _lib.check(lib().grapes_frontier_expand(_p(rowptr), _p(col), _p(nodes), nodes.numel(), _p(d_m), _p(eoff), e_cap,
                                            _p(src), _p(dst), _p(pos), _p(status), _stream()), "frontier_expand")
    return src, dst, pos


class _SliceRemarkArgs(__import__("ctypes").Structure):
    """include/grapes_hip.h: grapes_slice_remark_args"""
    _C = __import__("ctypes")
    _fields_ = [("mult", _C.c_void_p), ("unmark_ids", _C.c_void_p), ("n_unmark", _C.c_int32), ("d_n_unmark", _C.c_void_p),
                ("mark_ids", _C.c_void_p), ("n_mark", _C.c_int32), ("d_n_mark", _C.c_void_p), ("clear_bits", _C.c_void_p),
                ("clear_ids", _C.c_void_p), ("n_clear", _C.c_int32), ("d_n_clear", _C.c_void_p)]


def _remark_args(remark):
    """dict(mult=, unmark=(ids, d_n)|None, mark=..., clear=..., clear_bits=) -> (ctypes struct kept alive, byref) or (None, None)"""
    if remark is None:
        return None, None
    import ctypes as C
    _chk(remark.get("mult"), _i32, "mult", True); _chk(remark.get("clear_bits"), _i64, "clear_bits", True)
    a = _SliceRemarkArgs()
    a.mult = _p(remark.get("mult"))
    for key, f_ids, f_n, f_dn in (("unmark", "unmark_ids", "n_unmark", "d_n_unmark"), ("mark", "mark_ids", "n_mark", "d_n_mark"),
                                  ("clear", "clear_ids", "n_clear", "d_n_clear")):
        ids, dn = remark.get(key) or (None, None)
        _chk(ids, _i32, key, True)
        setattr(a, f_ids, _p(ids)); setattr(a, f_n, 0 if ids is None else ids.numel()); setattr(a, f_dn, _p(dn))
    a.clear_bits = _p(remark.get("clear_bits"))
    return a, C.byref(a)


class _DrawFinishArgs(__import__("ctypes").Structure):
    """include/grapes_hip.h: grapes_draw_finish_args (filled by grapes_gumbel_topk_deferred)"""
    _C = __import__("ctypes")
    _fields_ = [("parts_keys", _C.c_void_p), ("parts_emit", _C.c_void_p), ("sel", _C.c_void_p), ("keys_blocks", _C.c_int32),
                ("emit_block", _C.c_int32), ("n_host", _C.c_int32), ("d_n", _C.c_void_p), ("stats", _C.c_void_p),
                ("hist", _C.c_void_p), ("hist_words", _C.c_int32), ("stats_blocks", _C.c_int32)]


class _HopCountArgs(__import__("ctypes").Structure):
    """include/grapes_hip.h: grapes_hop_count_args"""
    _C = __import__("ctypes")
    _fields_ = [("indeg", _C.c_void_p), ("loops", _C.c_void_p), ("seginfo", _C.c_void_p), ("wsum", _C.c_void_p),
                ("slot", _C.c_void_p), ("n_long", _C.c_void_p)]


class _HopDegreeArgs(__import__("ctypes").Structure):
    """include/grapes_hip.h: grapes_hop_degree_args"""
    _C = __import__("ctypes")
    _fields_ = [("indeg", _C.c_void_p), ("loops", _C.c_void_p), ("seginfo", _C.c_void_p), ("wsum", _C.c_void_p),
                ("rowptr_t", _C.c_void_p), ("rowptr_s", _C.c_void_p), ("dinv", _C.c_void_p), ("seg_first", _C.c_void_p),
                ("row_loops", _C.c_void_p), ("long_items", _C.c_void_p), ("n_long", _C.c_void_p), ("item_cap", _C.c_int32),
                ("sync2", _C.c_void_p), ("cursor", _C.c_void_p)]


class HopCounters:
    """Per-graph counter tables of the counted hop build (include/grapes_hip.h: grapes_hop_count_args), indexed by GLOBAL node
    id, zero at rest and left zero by the launches that use them: in-degrees, self-loop counts, edge segments of the queried
    nodes, per-bitmap-word degree sums, and the second look-back scratch of the compaction."""

    def __init__(self, num_nodes, device):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("HopCounters: first use inside a stream capture; run one eager step first")
        W = (int(num_nodes) + 63) // 64
        self.num_nodes = int(num_nodes)
        self.indeg = torch.zeros(num_nodes, dtype=_i32, device=device)
        self.loops = torch.zeros(num_nodes, dtype=_i32, device=device)
        self.seginfo = torch.zeros(2 * num_nodes, dtype=_i32, device=device)
        self.wsum = torch.zeros(2 * W, dtype=_i32, device=device)
        self.sync2 = torch.zeros(256, dtype=_i64, device=device)


class HopBuild:
    """Arrays of ONE counted hop-graph build: the expansion fills `slot`, the compaction the row starts / dinv / segments, and
    PreparedGraph.counted() the CSRs and head records (two launches instead of grapes_gcn_prepare's four)."""

    def __init__(self, n_cap, e_cap, device, counters=None, cursor_form=None):
        self.n_cap, self.e_cap = int(n_cap), int(e_cap)
        # cursor form (A/B: GRAPES_HOP_CURSOR): the expansion's in-degree atomics return nothing and the fill takes an entry's place
        # from a per-row cursor the compaction wrote; else the atomic's return value IS the place (`slot`) and the fill has no atomic
        if cursor_form is None:
            cursor_form = _sw("GRAPES_HOP_CURSOR", "0") != "0"
        self.slot = None if cursor_form else torch.empty(max(e_cap, 1), dtype=_i32, device=device)
        self.cursor = torch.empty(max(n_cap, 1), dtype=_i32, device=device) if cursor_form else None
        self.rowptr_t = torch.empty(n_cap + 1, dtype=_i32, device=device)
        self.rowptr_s = torch.empty(n_cap + 1, dtype=_i32, device=device)
        self.dinv = torch.empty(max(n_cap, 1), dtype=_f32, device=device)
        self.seg_first = torch.empty(max(n_cap, 1), dtype=_i32, device=device)
        self.row_loops = torch.empty(max(n_cap, 1), dtype=_i32, device=device)
        self.csr_dst = torch.empty(max(e_cap, 1), dtype=_i32, device=device)      # (cleared by the compaction: zero= list)
        self.item_cap = int(lib().grapes_gcn_long_items_capacity(e_cap))
        self.long_items = torch.empty(4 * self.item_cap, dtype=_i32, device=device)
        self.n_long = counters if counters is not None else torch.empty(4, dtype=_i32, device=device)

    def count_args(self, hc: "HopCounters"):
        a = _HopCountArgs()
        a.indeg, a.loops, a.seginfo, a.wsum = _p(hc.indeg), _p(hc.loops), _p(hc.seginfo), _p(hc.wsum)
        a.slot, a.n_long = _p(self.slot), _p(self.n_long)
        return a

    def degree_args(self, hc: "HopCounters"):
        a = _HopDegreeArgs()
        a.indeg, a.loops, a.seginfo, a.wsum = _p(hc.indeg), _p(hc.loops), _p(hc.seginfo), _p(hc.wsum)
        a.rowptr_t, a.rowptr_s, a.dinv = _p(self.rowptr_t), _p(self.rowptr_s), _p(self.dinv)
        a.seg_first, a.row_loops = _p(self.seg_first), _p(self.row_loops)
        a.long_items, a.n_long, a.item_cap = _p(self.long_items), _p(self.n_long), self.item_cap
        a.sync2 = _p(hc.sync2)
        a.cursor = _p(self.cursor)
        return a


def slice_stage(e_cap, device):
    """Scratch for frontier_expand_fused(slice_stage=) -> PreparedGraph.small_batch(stages=): no clearing needed."""
    return torch.empty(int(lib().grapes_slice_stage_words(e_cap)), dtype=_i32, device=device)


def frontier_expand_fused(rowptr, col, nodes, e_cap, d_m=None, status=None, mark_prev_bits=None, mark_bits=None,
                          num_nodes=0, remark=None, count_mult=None, count_bsum=None, slice_stage=None, count=None, finish=None,
                          node_ext=None, node_ext_out=None):
    """frontier_offsets + frontier_expand in one launch (<= 2048 queried nodes): (src, dst, d_e, eoff).
    mark_bits (+ mark_prev_bits, num_nodes): also the hop's bitmap marks (bitmap_mark_hop) in the same launch.
    remark = dict(mult=, unmark=(ids, d_n)|None, mark=(ids, d_n)|None, clear=(ids, d_n)|None, clear_bits=): slice_remark in
    the same launch (clear_bits must not be mark_prev_bits).  count_mult + count_bsum: also slice_filter's counting half
    over the produced edges (count_bsum zeroed by the caller; pass it to slice_filter(bsum=...))."""
    _chk(rowptr, _i64, "rowptr"); _chk(col, _i32, "col"); _chk(nodes, _i32, "nodes")
    _chk(mark_prev_bits, _i64, "mark_prev_bits", True); _chk(mark_bits, _i64, "mark_bits", True)
    m, dev = nodes.numel(), nodes.device
    eoff = torch.empty(m + 1, dtype=_i32, device=dev)
    d_e = torch.empty(1, dtype=_i32, device=dev)
    src = torch.empty(e_cap, dtype=_i32, device=dev)
    dst = torch.empty(e_cap, dtype=_i32, device=dev)
    _keep, rm = _remark_args(remark)
    _chk(count_mult, _i32, "count_mult", True); _chk(count_bsum, _i32, "count_bsum", True); _chk(slice_stage, _i32, "slice_stage", True)
    if slice_stage is not None and slice_stage.numel() < int(lib().grapes_slice_stage_words(e_cap)):
        raise ValueError("frontier_expand_fused: slice_stage needs grapes_slice_stage_words(e_cap) words")
    # node_ext int64[2 m]: the queried rows' (rowptr[id], rowptr[id + 1]) written by the draw that produced `nodes`
    # (gumbel_topk(ext=...)["union_ext"]): one dependent round trip less; node_ext_out int64[2 m]: the extents this launch finds itself
    _chk(node_ext, _i64, "node_ext", True); _chk(node_ext_out, _i64, "node_ext_out", True)
    if (node_ext is not None and node_ext.numel() < 2 * m) or (node_ext_out is not None and node_ext_out.numel() < 2 * m):
        raise ValueError("frontier_expand_fused: node_ext / node_ext_out hold two int64 per queried node")
    if finish is not None or node_ext is not None or node_ext_out is not None or count is not None:
        # finish = gumbel_topk(defer_finish=True)["finish"]: one more workgroup ends that draw;
        # count = (HopCounters, HopBuild): the hop graph's degree counting rides in this launch
        import ctypes as C
        ca = None
        if count is not None:
            if count[1].e_cap < e_cap:
                raise ValueError("frontier_expand_fused: the HopBuild's slot array is smaller than e_cap")
            ca = count[1].count_args(count[0])
        _lib.check(lib().grapes_frontier_expand_fused_ext(_p(rowptr), _p(col), _p(nodes), m, _p(d_m), e_cap, _p(eoff), _p(d_e),
                                                          _p(src), _p(dst), _p(status), _p(mark_prev_bits), _p(mark_bits),
                                                          int(num_nodes), rm, _p(count_mult), _p(count_bsum), _p(slice_stage),
                                                          C.byref(ca) if ca is not None else None,
                                                          C.byref(finish[0]) if finish is not None else None,
                                                          _p(node_ext), _p(node_ext_out), _stream()),
                   "frontier_expand_fused_ext")
        return src, dst, d_e, eoff
    _lib.check(lib().grapes_frontier_expand_fused(_p(rowptr), _p(col), _p(nodes), m, _p(d_m), e_cap, _p(eoff), _p(d_e),
                                                  _p(src), _p(dst), _p(status), _p(mark_prev_bits), _p(mark_bits), int(num_nodes),
                                                  rm, _p(count_mult), _p(count_bsum), _p(slice_stage), _stream()), "frontier_expand_fused")
    return src, dst, d_e, eoff


# ------------------------------------------------------------------------------- bitmaps / compaction
def bitmap_mark(bits, bits1, ids, num_nodes, d_n=None, status=None):
    _chk(bits, _i64, "bits"); _chk(bits1, _i64, "bits1", True); _chk(ids, _i32, "ids")
    _lib.check(lib().grapes_bitmap_mark(_p(bits), _p(bits1), _p(ids), ids.numel(), _p(d_n), num_nodes, _p(status),
                                        _stream()), "bitmap_mark")


def bitmap_mark_rows(bits, bits1, nodes, eoff, num_nodes, d_m=None, status=None):
    """Marks the queried nodes that have >= 1 out-edge (the source endpoints of the frontier)."""
    _chk(bits, _i64, "bits"); _chk(bits1, _i64, "bits1", True); _chk(nodes, _i32, "nodes"); _chk(eoff, _i32, "eoff")
    _lib.check(lib().grapes_bitmap_mark_rows(_p(bits), _p(bits1), _p(nodes), nodes.numel(), _p(d_m), _p(eoff), num_nodes,
                                             _p(status), _stream()), "bitmap_mark_rows")


def bitmap_clear(bits, ids, d_n=None):
    _chk(bits, _i64, "bits"); _chk(ids, _i32, "ids")
    _lib.check(lib().grapes_bitmap_clear(_p(bits), _p(ids), ids.numel(), _p(d_n), _stream()), "bitmap_clear")


def bitmap_mark_hop(prev_bits, bits, bits1, previous, eoff, dst, num_nodes, d_m=None, d_e=None, status=None):
    """The three marks of a hop in one launch (== bitmap_mark(prev) + bitmap_mark_rows + bitmap_mark(dst))."""
    _chk(prev_bits, _i64, "prev_bits"); _chk(bits, _i64, "bits"); _chk(bits1, _i64, "bits1", True)
    _chk(previous, _i32, "previous"); _chk(eoff, _i32, "eoff"); _chk(dst, _i32, "dst")
    _lib.check(lib().grapes_bitmap_mark_hop(_p(prev_bits), _p(bits), _p(bits1), _p(previous), previous.numel(), _p(d_m),
                                            _p(eoff), _p(dst), dst.numel(), _p(d_e), num_nodes, _p(status), _stream()),
               "bitmap_mark_hop")


def bitmap_mark_lists(bits, bits1, lists, num_nodes, status=None, unmark_mult=None):
    """lists: up to four (ids, d_n or None) pairs marked in one launch.  unmark_mult: the slice multiplicity table, zeroed
    at every listed id in the same launch (see slice_remark)."""
    _chk(bits, _i64, "bits"); _chk(bits1, _i64, "bits1", True); _chk(unmark_mult, _i32, "unmark_mult", True)
    if len(lists) > 4:
        raise ValueError("at most four lists per launch")
    args = []
    for ids, d_n in list(lists) + [(None, None)] * (4 - len(lists)):
        _chk(ids, _i32, "ids", True)
        args += [_p(ids), 0 if ids is None else ids.numel(), _p(d_n)]
    _lib.check(lib().grapes_bitmap_mark_lists(_p(bits), _p(bits1), *args, num_nodes, _p(status), _p(unmark_mult), _stream()),
               "bitmap_mark_lists")


def union_sorted(lists, num_nodes, n_cap, node_map=None, status=None, unmark_mult=None):
    """Ascending duplicate-free union of up to four (ids, d_n or None) lists with at most 4096 ids in all, by one workgroup:
    (all_ids[n_cap], counts[2]); node_map[id] = rank; unmark_mult zeroed at the ids."""
    if not (1 <= len(lists) <= 4) or sum(ids.numel() for ids, _ in lists) > 4096:
        raise ValueError("union_sorted: 1..4 lists, at most 4096 ids in all")
    _chk(node_map, _i32, "node_map", True); _chk(unmark_mult, _i32, "unmark_mult", True)
    dev = lists[0][0].device
    args = []
    for ids, d_n in list(lists) + [(None, None)] * (4 - len(lists)):
        _chk(ids, _i32, "ids", True)
        args += [_p(ids), 0 if ids is None else ids.numel(), _p(d_n)]
    out = torch.empty(n_cap, dtype=_i32, device=dev)
    counts = torch.empty(2, dtype=_i32, device=dev)
    _lib.check(lib().grapes_union_sorted(*args, num_nodes, n_cap, _p(out), _p(node_map), _p(counts), _p(unmark_mult), _p(status),
                                         _stream()), "union_sorted")
    return out, counts


def frontier_compact(bits, bits1, prev_bits, num_nodes, n_cap, node_map=None, status=None, ind_code=None, epoch=0,
                     d_epoch=None, ind_bit=0, sync=None, one_launch=True, want_cand_pos=False, zero=(), remark=None,
                     degrees=None):
    """Returns (batch_nodes[n_cap], neighbor_nodes[n_cap], nb_local[n_cap], counts[2]) — ascending ids.
    ind_code: also set indicator bit `ind_bit` of every emitted neighbour (main.py:191)."""
    _chk(ind_code, _i32, "ind_code", True)
    _chk(bits, _i64, "bits"); _chk(bits1, _i64, "bits1", True); _chk(prev_bits, _i64, "prev_bits", True)
    _chk(node_map, _i32, "node_map", True)
    dev = bits.device
    batch = torch.empty(n_cap, dtype=_i32, device=dev)
    neigh = torch.empty(n_cap, dtype=_i32, device=dev)
    nbl = torch.empty(n_cap, dtype=_i32, device=dev)
    counts = torch.empty(2, dtype=_i32, device=dev)
    ws = _ws(lib().grapes_frontier_compact_workspace_bytes(n_cap, num_nodes), dev)
    if sync is None and one_launch:
        sync = sync_scratch(dev)
    _chk(sync, _i64, "sync", True)
    cand_pos = torch.empty(n_cap, dtype=_i32, device=dev) if want_cand_pos else None
    # zero: up to three (tensor, words) pairs the launch also clears (scratch of the operation that follows)
    _keep, crm = _remark_args(remark)
    zargs = []
    for zt, zw in list(zero)[:3] + [(None, 0)] * (3 - len(list(zero)[:3])):
        zargs += [_p(zt), int(zw)]
    if degrees is not None:    # degrees = (HopCounters, HopBuild): row starts, dinv and segments of the hop graph from this launch
        import ctypes as C
        if degrees[1].n_cap < n_cap:
            raise ValueError("frontier_compact: the HopBuild is smaller than n_cap")
        da = degrees[1].degree_args(degrees[0])
        _lib.check(lib().grapes_frontier_compact_counted(_p(bits), _p(bits1), _p(prev_bits), num_nodes, n_cap, _p(batch), _p(neigh),
                                                         _p(nbl), _p(node_map), _p(counts), _p(ind_code), epoch, _p(d_epoch),
                                                         ind_bit, _p(cand_pos), *zargs, crm, _p(ws), _p(sync), _p(status),
                                                         C.byref(da), _stream()), "frontier_compact_counted")
    else:
        _lib.check(lib().grapes_frontier_compact(_p(bits), _p(bits1), _p(prev_bits), num_nodes, n_cap, _p(batch), _p(neigh),
                                                 _p(nbl), _p(node_map), _p(counts), _p(ind_code), epoch, _p(d_epoch), ind_bit,
                                                 _p(cand_pos), *zargs, crm, _p(ws), _p(sync), _p(status), _stream()),
                   "frontier_compact")
    if want_cand_pos:
        return batch, neigh, nbl, counts, cand_pos
    return batch, neigh, nbl, counts


# ------------------------------------------------------------------------------- slice
def slice_mark(mult, cols, unmark=False, d_c=None, clear_bits=None):
    _chk(mult, _i32, "mult"); _chk(cols, _i32, "cols"); _chk(clear_bits, _i64, "clear_bits", True)
    _lib.check(lib().grapes_slice_mark(_p(mult), _p(cols), cols.numel(), _p(d_c), 1 if unmark else 0, _p(clear_bits),
                                       _stream()), "slice_mark")


def slice_remark(mult, unmark=None, mark=None, clear=None, clear_bits=None):
    """unmark / mark / clear: (ids, d_n or None) or None.  One launch: zero mult at `unmark`, +1 at `mark` (the two lists
    must be disjoint), zero the words of `clear_bits` holding `clear`."""
    _chk(mult, _i32, "mult"); _chk(clear_bits, _i64, "clear_bits", True)
    args = []
    for pair in (unmark, mark):
        ids, d_n = pair if pair is not None else (None, None)
        _chk(ids, _i32, "ids", True)
        args += [_p(ids), 0 if ids is None else ids.numel(), _p(d_n)]
    ids, d_n = clear if clear is not None else (None, None)
    _chk(ids, _i32, "clear ids", True)
    _lib.check(lib().grapes_slice_remark(_p(mult), *args, _p(clear_bits), _p(ids), 0 if ids is None else ids.numel(), _p(d_n),
                                         _stream()), "slice_remark")


def slice_filter_bsum(e_cap, device):
    """Zeroed per-1024-edge-block survivor counters for frontier_expand_fused(count_bsum=) + slice_filter(bsum=)."""
    return torch.zeros(max(int(lib().grapes_slice_filter_workspace_bytes(e_cap)) // 4, 1), dtype=_i32, device=device)


def slice_filter(mult, src, dst, out_cap, d_e=None, status=None, one_launch=None, bsum=None):
    """one_launch: the look-back form (default: GRAPES_ONE_LAUNCH_SLICE, off — 7 us/step slower than two launches).
    bsum: survivor counts already formed by the expansion that produced src / dst (only the emitting launch runs)."""
    if one_launch is None:
        one_launch = _ONE_LAUNCH_SLICE
    _chk(mult, _i32, "mult"); _chk(src, _i32, "src"); _chk(dst, _i32, "dst")
    dev = src.device
    out_src = torch.empty(out_cap, dtype=_i32, device=dev)
    out_dst = torch.empty(out_cap, dtype=_i32, device=dev)
    cnt = torch.empty(1, dtype=_i32, device=dev)
    _chk(bsum, _i32, "bsum", True)
    ws = bsum if bsum is not None else _ws(lib().grapes_slice_filter_workspace_bytes(src.numel()), dev)
    _lib.check(lib().grapes_slice_filter(_p(mult), _p(src), _p(dst), src.numel(), _p(d_e), out_cap, _p(out_src),
                                         _p(out_dst), _p(cnt), _p(ws), _p(sync_scratch(dev)) if one_launch else None,
                                         1 if bsum is not None else 0, _p(status), _stream()), "slice_filter")
    return out_src, out_dst, cnt


# ------------------------------------------------------------------------------- features
def indicator_mark(ind_code, ids, epoch, bit, d_n=None, d_epoch=None, advance_epoch=False):
    """advance_epoch: a new batch — marks with *d_epoch + 1 and stores that back (no separate counter launch)."""
    _chk(ind_code, _i32, "ind_code"); _chk(ids, _i32, "ids")
    _lib.check(lib().grapes_indicator_mark(_p(ind_code), _p(ids), ids.numel(), _p(d_n), epoch, _p(d_epoch), bit,
                                           1 if advance_epoch else 0, _stream()), "indicator_mark")


def step_begin(ids32, d_cursor, stride, offset, targets, ind_code=None, d_epoch=None, bit=0, counters=None, totals=None):
    """First launch of a self-feeding captured step: next batch of target ids from the device-resident id array, cursor and
    epoch advanced, target indicators marked, the previous step's edge counters (counters[:, col] view) added to totals."""
    _chk(ids32, _i32, "ids"); _chk(d_cursor, _i32, "d_cursor"); _chk(targets, _i32, "targets")
    _chk(ind_code, _i32, "ind_code", True); _chk(d_epoch, _i32, "d_epoch", True); _chk(totals, _i64, "totals", True)
    cs, nc = (int(counters.stride(0)), int(counters.numel())) if counters is not None else (1, 0)
    _lib.check(lib().grapes_step_begin(_p(ind_code), _p(d_epoch), bit, _p(ids32), ids32.numel(), _p(d_cursor), int(stride),
                                       int(offset), targets.numel(), _p(targets), _p(counters), cs, nc, _p(totals), _stream()),
               "step_begin")


def gather_rows(X, ids, ind_code=None, epoch=0, num_ind=0, d_n=None, out=None, d_epoch=None):
    _chk(X, _f32, "X"); _chk(ids, _i32, "ids"); _chk(ind_code, _i32, "ind_code", True)
    n, F = ids.numel(), X.shape[1]
    if out is None:
        out = torch.empty((n, F + num_ind), dtype=_f32, device=X.device)
    _lib.check(lib().grapes_gather_rows(_p(X), F, _p(ids), n, _p(d_n), _p(ind_code), epoch, _p(d_epoch), num_ind, _p(out),
                                        _stream()),
               "gather_rows")
    return out


# ------------------------------------------------------------------------------- GCN
_SMALL_GRAPH = 2048      # <= this many nodes (the classifier's sampled subgraphs): no split of long rows into work items


class PreparedGraph:
    """gcn_norm + CSR by target / by source of one (local) edge list."""

    __slots__ = ("n", "e", "d_n", "d_e", "rowptr_t", "csr_src", "rowptr_s", "csr_dst", "dinv", "status",
                 "long_items", "n_long", "item_cap", "items_t", "items_s", "n_items_t", "n_items_s",
                 "items_fwd", "row_head", "head_ids", "head_local")

    @staticmethod
    def scratch(n, e, device):
        """(workspace, csr_dst, [(tensor, words), ...]): the build's scratch allocated ahead of it, with the ranges an earlier
        launch (frontier_compact(zero=...)) has to clear for `prezeroed=True`."""
        ws = _ws(lib().grapes_gcn_prepare_workspace_bytes(n, e), device)
        csr_dst = torch.empty(max(e, 1), dtype=_i32, device=device)
        return ws, csr_dst, [(ws, lib().grapes_gcn_prepare_zero_words(n)), (csr_dst, e)]

    def __init__(self, edge_src, edge_dst, n, d_n=None, d_e=None, status=None, src_grouped=False, items_fwd=True,
                 node_map=None, head_ids=None, counters=None, scratch=None, prefetch=None, head_local=False):
        """node_map: edge_src / edge_dst are GLOBAL ids, relabelled through this table inside the build.
        head_local: head_ids is 0, 1, 2, ... — the head records then hold LOCAL ids and drive the record form of the
        aggregation over [n, f] activations (gcn_aggregate_fwd / _fwd_head -> grapes_gcn_aggregate_fwd_rec).
        head_ids: int32[n] feature-matrix row of every local node (the hop's batch_nodes): the build also writes the
        per-row head records the fused gather-SpMM (gcn_aggregate_gather) reads.
        counters: int32[4] to use for the build's device counters ([2] = aggregated edges), e.g. a row of a caller's table."""
        _chk(head_ids, _i32, "head_ids", True); _chk(counters, _i32, "counters", True)
        _chk(edge_src, _i32, "edge_src"); _chk(edge_dst, _i32, "edge_dst"); _chk(node_map, _i32, "node_map", True)
        dev = edge_src.device
        e = edge_src.numel()
        self.n, self.e, self.d_n, self.d_e, self.status = n, e, d_n, d_e, status
        # items_fwd=False: the forward (by-target) aggregation runs as ONE launch, every row by one wavefront —
        # right for frontier graphs, whose in-degrees are tiny (<= number of source rows); still correct otherwise
        self.items_fwd = items_fwd
        self.rowptr_t = torch.empty(n + 1, dtype=_i32, device=dev)
        self.rowptr_s = torch.empty(n + 1, dtype=_i32, device=dev)
        self.csr_src = torch.empty(max(e, 1), dtype=_i32, device=dev)
        self.csr_dst = scratch[1] if scratch is not None else torch.empty(max(e, 1), dtype=_i32, device=dev)
        self.dinv = torch.empty(max(n, 1), dtype=_f32, device=dev)
        self.item_cap = lib().grapes_gcn_long_items_capacity(e)
        self.long_items = torch.empty(4 * self.item_cap, dtype=_i32, device=dev)
        self.n_long = counters if counters is not None else torch.empty(4, dtype=_i32, device=dev)
        self.items_t, self.items_s = self.long_items[: 2 * self.item_cap], self.long_items[2 * self.item_cap:]
        self.n_items_t, self.n_items_s = self.n_long[0:1], self.n_long[1:2]
        self.head_ids = head_ids
        self.head_local = bool(head_local) and head_ids is not None
        self.row_head = torch.empty((max(n, 1), 12), dtype=_i32, device=dev) if (head_ids is not None and e > 0) else None
        ws = scratch[0] if scratch is not None else _ws(lib().grapes_gcn_prepare_workspace_bytes(n, e), dev)
        flags = (1 if src_grouped else 0) | (2 if scratch is not None else 0)
        args = (_p(edge_src), _p(edge_dst), e, _p(d_e), _p(node_map), n, _p(d_n), flags,
                _p(self.rowptr_t), _p(self.csr_src), _p(self.rowptr_s), _p(self.csr_dst),
                _p(self.dinv), _p(self.long_items), _p(self.n_long),
                _p(head_ids) if self.row_head is not None else None, _p(self.row_head),
                _p(ws), _p(sync_scratch(dev)) if _ONE_LAUNCH_PREP else None, _p(status))
        if prefetch is not None and head_ids is not None and _PREFETCH_ROWS:
            # prefetch = (X, row_floats): the build's first launch also touches X[head_ids] (the gather-SpMM's rows; see the header)
            _lib.check(lib().grapes_gcn_prepare_prefetching(*args, _p(prefetch[0]), int(prefetch[0].stride(0)), int(prefetch[1]),
                                                            _stream()), "gcn_prepare_prefetching")
        else:
            _lib.check(lib().grapes_gcn_prepare(*args, _stream()), "gcn_prepare")

    @classmethod
    def counted(cls, edge_src, edge_dst, hb: "HopBuild", n, d_n, d_e, node_map, status=None, head_ids=None, prefetch=None,
                head_local=False):
        """The hop graph from a COUNTED expansion + compaction (frontier_expand_fused(count=), frontier_compact(degrees=) over the
        same HopBuild): the two remaining launches of the build.  Same arrays as PreparedGraph(src_grouped=True, node_map=...)."""
        _chk(edge_src, _i32, "edge_src"); _chk(edge_dst, _i32, "edge_dst"); _chk(node_map, _i32, "node_map"); _chk(head_ids, _i32, "head_ids", True)
        dev = edge_src.device
        e = edge_src.numel()
        g = object.__new__(cls)
        g.n, g.e, g.d_n, g.d_e, g.status, g.items_fwd = n, e, d_n, d_e, status, False
        g.rowptr_t, g.rowptr_s, g.dinv, g.csr_dst = hb.rowptr_t, hb.rowptr_s, hb.dinv, hb.csr_dst
        g.csr_src = torch.empty(max(e, 1), dtype=_i32, device=dev)
        g.item_cap, g.long_items, g.n_long = hb.item_cap, hb.long_items, hb.n_long
        g.items_t, g.items_s = g.long_items[: 2 * g.item_cap], g.long_items[2 * g.item_cap:]
        g.n_items_t, g.n_items_s = g.n_long[0:1], g.n_long[1:2]
        g.head_ids = head_ids
        g.head_local = bool(head_local) and head_ids is not None
        g.row_head = torch.empty((max(n, 1), 12), dtype=_i32, device=dev) if head_ids is not None else None
        tmp = _tmp(torch.empty(max(e, 1), dtype=_i32, device=dev))
        pf = prefetch if (prefetch is not None and head_ids is not None and _PREFETCH_ROWS) else None
        _lib.check(lib().grapes_gcn_prepare_counted(_p(edge_src), _p(edge_dst), _p(hb.slot), e, _p(d_e), _p(node_map), n, _p(d_n),
                                                    _p(hb.rowptr_t), _p(hb.rowptr_s), _p(hb.seg_first), _p(hb.row_loops), _p(hb.dinv),
                                                    _p(g.csr_src), _p(g.csr_dst), _p(tmp), _p(head_ids), _p(g.row_head), _p(status),
                                                    _p(pf[0]) if pf else None, int(pf[0].stride(0)) if pf else 0,
                                                    int(pf[1]) if pf else 0, _p(hb.cursor), _stream()), "gcn_prepare_counted")
        return g

    @classmethod
    def small_batch(cls, edge_lists, n, d_n=None, status=None, node_map=None, head_ids=None, counters=None, stages=None):
        """Several graphs over the SAME n <= 2048 nodes in one launch (one workgroup each): edge_lists =
        [(edge_src, edge_dst, d_e), ...] in frontier (source-grouped) order.  Returns the PreparedGraphs.
        stages: per graph None or (slice_stage, d_fe, fe_cap) — the edge list (edge_src / edge_dst / d_e: outputs then) is
        first assembled from the stage a frontier_expand_fused(slice_stage=) launch left (slice_adjacency without a launch)."""
        import ctypes as C
        if not (1 <= len(edge_lists) <= 8) or n > _SMALL_GRAPH:
            raise ValueError("small_batch: 1..8 graphs of at most %d nodes" % _SMALL_GRAPH)
        outs, wss = [], []
        for gi, (es, ed, d_e) in enumerate(edge_lists):
            _chk(es, _i32, "edge_src"); _chk(ed, _i32, "edge_dst")
            dev, e = es.device, es.numel()
            g = object.__new__(cls)
            g.n, g.e, g.d_n, g.d_e, g.status, g.items_fwd = n, e, d_n, d_e, status, True
            g.rowptr_t = torch.empty(n + 1, dtype=_i32, device=dev); g.rowptr_s = torch.empty(n + 1, dtype=_i32, device=dev)
            g.csr_src = torch.empty(max(e, 1), dtype=_i32, device=dev); g.csr_dst = torch.empty(max(e, 1), dtype=_i32, device=dev)
            g.dinv = torch.empty(max(n, 1), dtype=_f32, device=dev)
            g.item_cap = lib().grapes_gcn_long_items_capacity(e)
            g.long_items = torch.empty(4 * g.item_cap, dtype=_i32, device=dev)
            g.n_long = counters[gi] if counters is not None else torch.empty(4, dtype=_i32, device=dev)
            g.items_t, g.items_s = g.long_items[: 2 * g.item_cap], g.long_items[2 * g.item_cap:]
            g.n_items_t, g.n_items_s = g.n_long[0:1], g.n_long[1:2]
            g.head_ids = head_ids
            g.head_local = False
            g.row_head = torch.empty((max(n, 1), 12), dtype=_i32, device=dev) if head_ids is not None else None
            outs.append(g)
            wss.append(_ws(lib().grapes_gcn_prepare_workspace_bytes(n, e), dev))
        k = len(outs)
        arr = lambda ts: (C.c_void_p * k)(*[None if t is None else t.data_ptr() for t in ts])
        ecnt = (C.c_int32 * k)(*[g.e for g in outs])
        _lib.check(lib().grapes_gcn_prepare_small_batch(
            k, arr([el[0] for el in edge_lists]), arr([el[1] for el in edge_lists]), ecnt, arr([el[2] for el in edge_lists]),
            _p(node_map), n, _p(d_n), arr([g.rowptr_t for g in outs]), arr([g.csr_src for g in outs]),
            arr([g.rowptr_s for g in outs]), arr([g.csr_dst for g in outs]), arr([g.dinv for g in outs]),
            arr([g.long_items for g in outs]), arr([g.n_long for g in outs]), _p(head_ids),
            arr([g.row_head for g in outs]), arr(wss),
            arr([None if st is None else st[0] for st in stages]) if stages is not None else None,
            arr([None if st is None else st[1] for st in stages]) if stages is not None else None,
            (C.c_int32 * k)(*[0 if st is None else int(st[2]) for st in stages]) if stages is not None else None,
            _p(status), _stream()), "gcn_prepare_small_batch")
        return outs

    @classmethod
    def from_csr(cls, rowptr_t32, csr_src, n, rowptr_s32=None, csr_dst=None):
        """Full-graph form: an int32 CSR by target (ascending columns, no self-loops) and, for the backward
        pass, optionally the CSR by source (for a symmetric graph the same arrays)."""
        _chk(rowptr_t32, _i32, "rowptr_t"); _chk(csr_src, _i32, "csr_src")
        dev = rowptr_t32.device
        self = object.__new__(cls)
        e = csr_src.numel()
        self.n, self.e, self.d_n, self.d_e, self.status, self.items_fwd = n, e, None, None, None, True
        self.row_head = self.head_ids = None
        self.head_local = False
        self.rowptr_t, self.csr_src = rowptr_t32, csr_src
        self.rowptr_s, self.csr_dst = (rowptr_s32, csr_dst) if rowptr_s32 is not None else (rowptr_t32, csr_src)
        self.dinv = torch.empty(max(n, 1), dtype=_f32, device=dev)
        self.item_cap = lib().grapes_gcn_long_items_capacity(e)
        self.long_items = torch.empty(4 * self.item_cap, dtype=_i32, device=dev)
        self.n_long = torch.zeros(4, dtype=_i32, device=dev)
        self.n_long[2] = e
        self.items_t, self.items_s = self.long_items[: 2 * self.item_cap], self.long_items[2 * self.item_cap:]
        self.n_items_t, self.n_items_s = self.n_long[0:1], self.n_long[1:2]
        _lib.check(lib().grapes_gcn_prepare_from_csr(_p(self.rowptr_t), n, _p(self.dinv), _p(self.items_t),
                                                     _p(self.n_items_t), self.item_cap, _stream()), "gcn_prepare_from_csr")
        if rowptr_s32 is None:      # symmetric: the by-source items are the by-target items
            self.items_s, self.n_items_s = self.items_t, self.n_items_t
        else:
            tmp = torch.empty_like(self.dinv)
            _lib.check(lib().grapes_gcn_prepare_from_csr(_p(self.rowptr_s), n, _p(tmp), _p(self.items_s),
                                                         _p(self.n_items_s), self.item_cap, _stream()), "gcn_prepare_from_csr")
        return self

    @property
    def num_edges_no_loops(self) -> torch.Tensor:
        """1-element device tensor: number of aggregated (non-self-loop) edges (also valid for capacity-padded
        graphs whose true sizes live on the device)."""
        return self.n_long[2:3]


def linear_fwd(x, w, d_n=None, out=None):
    _chk(x, _f32, "x"); _chk(w, _f32, "w")
    n, fi = x.shape
    fo = w.shape[0]
    if w.shape[1] != fi:
        raise ValueError("weight / input width mismatch")
    if out is None:
        out = torch.empty((n, fo), dtype=_f32, device=x.device)
    _lib.check(lib().grapes_linear_fwd(_p(x), _p(w), _p(out), n, _p(d_n), fi, fo, _stream()), "linear_fwd")
    return out


def eval_predict(logits, node_map, targets, status=None, want_rows=True):
    """eval.py:154-155: (argmax of the targets' logit rows [B] int64, the rows [B, C] or None) in one launch."""
    _chk(logits, _f32, "logits"); _chk(node_map, _i32, "node_map"); _chk(targets, _i32, "targets"); _chk(status, _i32, "status", True)
    n, c = logits.shape
    b = targets.numel()
    pred = torch.empty(b, dtype=_i64, device=logits.device)
    rows = torch.empty((b, c), dtype=_f32, device=logits.device) if want_rows else None
    _lib.check(lib().grapes_eval_predict(_p(logits), n, c, _p(node_map), _p(targets), b, _p(pred), _p(rows), _p(status), _stream()),
               "eval_predict")
    return pred, rows


def linear_fwd_row_scaled(x, w, row_scale, d_n=None, out=None):
    """diag(row_scale) · x wᵀ in one launch (the full-batch inference transform: rows leave the GEMM scaled by dinv) — the same bits
    as scale_rows(linear_fwd(x, w), row_scale)."""
    _chk(x, _f32, "x"); _chk(w, _f32, "w"); _chk(row_scale, _f32, "row_scale")
    n, fi = x.shape
    fo = w.shape[0]
    if w.shape[1] != fi or row_scale.numel() < n:
        raise ValueError("weight / input width or scale length mismatch")
    if out is None:
        out = torch.empty((n, fo), dtype=_f32, device=x.device)
    _lib.check(lib().grapes_linear_fwd_row_scaled(_p(x), _p(w), _p(row_scale), _p(out), n, _p(d_n), fi, fo, _stream()),
               "linear_fwd_row_scaled")
    return out


def linear_bwd_weight(dh, x, d_n=None, out=None, accumulate=False, defer=None):
    _chk(dh, _f32, "dh"); _chk(x, _f32, "x")
    n, fi = x.shape
    fo = dh.shape[1]
    if out is None:
        out = torch.empty((fo, fi), dtype=_f32, device=x.device)
        accumulate = False
    if defer is not None and fo > 1 and defer.try_add(dh, None, x, d_n, out, None, accumulate):
        return out
    ws = _ws(lib().grapes_linear_bwd_weight_workspace_bytes(n, fi, fo), x.device)
    _lib.check(lib().grapes_linear_bwd_weight(_p(dh), _p(x), _p(out), n, _p(d_n), fi, fo, 1 if accumulate else 0, _p(ws),
                                              _stream()), "linear_bwd_weight")
    return out


def linear_bwd_weight_and_input(dh, x, w, d_n=None, out=None, accumulate=False, defer=None):
    """(dW = dhᵀ x, dx = dh w) of one linear map (modules/gcn.py:32,36 backward).  With defer (a DeferredSlabs) and shapes of the
    two few-row kernels both run as ONE launch; otherwise the two separate entry points.  -> dx"""
    _chk(dh, _f32, "dh"); _chk(x, _f32, "x"); _chk(w, _f32, "w")
    n, fi = x.shape
    fo = dh.shape[1]
    if defer is not None and fo > 1 and out is not None:
        dx = torch.empty((n, fi), dtype=_f32, device=dh.device)
        if defer.try_add(dh, None, x, d_n, out, None, accumulate, w=w, dx=dx):
            return dx
    linear_bwd_weight(dh, x, d_n=d_n, out=out, accumulate=accumulate, defer=defer)
    return linear_bwd_input(dh, w, d_n=d_n)


def linear_bias_act_fwd(x, w, bias=None, relu=False, d_n=None, out=None):
    """act(x Wᵀ + b) in one GEMM (aggregate-first layers)."""
    _chk(x, _f32, "x"); _chk(w, _f32, "w"); _chk(bias, _f32, "bias", True)
    n, fi = x.shape
    fo = w.shape[0]
    if out is None:
        out = torch.empty((n, fo), dtype=_f32, device=x.device)
    _lib.check(lib().grapes_linear_bias_act_fwd(_p(x), _p(w), _p(bias), 1 if relu else 0, _p(out), n, _p(d_n), fi, fo,
                                                _stream()), "linear_bias_act_fwd")
    return out


def linear_bias_act_head_fwd(x, w, bias, relu, head_w, d_n=None):
    """(act(x Wᵀ + b), its product with the 1-wide head weight head_w [1, F_out] or [F_out]) — one launch where the
    bf16x3 GEMM applies."""
    _chk(x, _f32, "x"); _chk(w, _f32, "w"); _chk(bias, _f32, "bias", True); _chk(head_w, _f32, "head_w")
    n, fi = x.shape
    fo = w.shape[0]
    if head_w.numel() != fo:
        raise ValueError("linear_bias_act_head_fwd: head_w must have F_out elements")
    out = torch.empty((n, fo), dtype=_f32, device=x.device)
    head = torch.empty((n, 1), dtype=_f32, device=x.device)
    _lib.check(lib().grapes_linear_bias_act_head_fwd(_p(x), _p(w), _p(bias), 1 if relu else 0, _p(out), _p(head_w), _p(head),
                                                     n, _p(d_n), fi, fo, _stream()), "linear_bias_act_head_fwd")
    return out, head


def split_gemm_available(n, f_in, f_out) -> bool:
    """True where the bf16x3 forward / dW kernels (and with them the strided-input forms below) apply."""
    return bool(lib().grapes_split_gemm_available(int(n), int(f_in), int(f_out)))


def _row_strided(x, f_in, name):
    """x: fp32 [n, >= f_in] whose rows may be a column-slice view of a wider contiguous matrix."""
    if x.dtype != _f32 or not x.is_cuda or x.dim() != 2 or x.stride(1) != 1 or x.shape[1] != f_in:
        raise _lib.GrapesHipError(f"{name}: expected a CUDA fp32 [n, {f_in}] matrix with unit column stride")
    return int(x.stride(0))


def linear_relu_head_fwd_bits_pair(x, w, bias, head_w, x_b, w_b, bias_b, head_w_b, d_n=None):
    """Two linear_relu_head_fwd_bits over the SAME rows in one launch (the sampler net's and the log-Z net's first layers at hop 0):
    -> (GateBits, head, GateBits_b, head_b), or None where the pair kernel does not take the shapes (call them separately)."""
    for t, nm in ((w, "w"), (bias, "bias"), (head_w, "head_w"), (w_b, "w_b"), (bias_b, "bias_b"), (head_w_b, "head_w_b")):
        _chk(t, _f32, nm)
    n, fi = x.shape
    fib = x_b.shape[1]
    fo = w.shape[0]
    if x_b.shape[0] != n or w_b.shape[0] != fo:
        return None
    ldx, ldxb = _row_strided(x, fi, "linear_relu_head_fwd_bits_pair"), _row_strided(x_b, fib, "linear_relu_head_fwd_bits_pair")
    nw = (fo + 31) // 32
    words = torch.empty((n, nw), dtype=torch.int32, device=x.device); head = torch.empty((n, 1), dtype=_f32, device=x.device)
    words_b = torch.empty((n, nw), dtype=torch.int32, device=x.device); head_b = torch.empty((n, 1), dtype=_f32, device=x.device)
    rc = lib().grapes_linear_relu_head_fwd_bits_pair(x.data_ptr(), ldx, _p(w), _p(bias), _p(head_w), _p(words), _p(head),
                                                     x_b.data_ptr(), ldxb, _p(w_b), _p(bias_b), _p(head_w_b), _p(words_b), _p(head_b),
                                                     fib, n, _p(d_n), fi, fo, _stream())
    if rc == -1:
        return None
    _lib.check(rc, "linear_relu_head_fwd_bits_pair")
    return GateBits(words, n, fo), head, GateBits(words_b, n, fo), head_b


def linear_bias_act_head_fwd_strided(x, w, bias, relu, head_w, d_n=None):
    """linear_bias_act_head_fwd for an x whose rows are x.stride(0) floats apart (a leading-columns view of a wider
    matrix).  Only where split_gemm_available(...)."""
    _chk(w, _f32, "w"); _chk(bias, _f32, "bias", True); _chk(head_w, _f32, "head_w", True)
    n, fi = x.shape
    fo = w.shape[0]
    ldx = _row_strided(x, fi, "linear_bias_act_head_fwd_strided")
    out = torch.empty((n, fo), dtype=_f32, device=x.device)
    head = torch.empty((n, 1), dtype=_f32, device=x.device) if head_w is not None else None
    _lib.check(lib().grapes_linear_bias_act_head_fwd_strided(x.data_ptr(), ldx, _p(w), _p(bias), 1 if relu else 0, _p(out),
                                                             _p(head_w), _p(head), n, _p(d_n), fi, fo, _stream()),
               "linear_bias_act_head_fwd_strided")
    return out, head


def linear_bwd_weight_gated_strided(x, gate, row_scale, col_vec, dw, dbias=None, dw_head=None, d_n=None, accumulate=False):
    """Rank-1 gated dW (see linear_bwd_weight_gated) for a row-strided x."""
    _chk(gate, _f32, "gate"); _chk(row_scale, _f32, "row_scale"); _chk(col_vec, _f32, "col_vec"); _chk(dw, _f32, "dw")
    _chk(dbias, _f32, "dbias", True); _chk(dw_head, _f32, "dw_head", True)
    n, fi = x.shape
    fo = gate.shape[1]
    ldx = _row_strided(x, fi, "linear_bwd_weight_gated_strided")
    ws = _ws(lib().grapes_linear_bwd_weight_gated_workspace_bytes(n, fi, fo), x.device)
    _lib.check(lib().grapes_linear_bwd_weight_gated_strided(_p(gate), x.data_ptr(), ldx, _p(row_scale), n, _p(d_n), _p(col_vec),
                                                            _p(dw), _p(dbias), _p(dw_head), fi, fo, 1 if accumulate else 0,
                                                            _p(ws), _stream()), "linear_bwd_weight_gated_strided")


def linear_bwd_weight_gated(dout, x, gate=None, d_n=None, dw=None, dbias=None, accumulate=False, want_bias=True,
                            row_scale=None, col_vec=None, dw_head=None, defer=None):
    """dW (+)= (dout ⊙ [gate>0])ᵀ x and dbias (+)= colsum(dout ⊙ [gate>0]) in one split-K GEMM.
    dw_head (rank-1 mode): also dw_head (+)= row_scaleᵀ·gate, the 1-wide head's weight gradient."""
    _chk(dw_head, _f32, "dw_head", True)
    _chk(dout, _f32, "dout", row_scale is not None); _chk(x, _f32, "x"); _chk(gate, _f32, "gate", True)
    _chk(row_scale, _f32, "row_scale", True); _chk(col_vec, _f32, "col_vec", True)
    n, fi = x.shape
    fo = dout.shape[1] if dout is not None else gate.shape[1]
    dev = x.device
    if dw is None:
        dw = torch.empty((fo, fi), dtype=_f32, device=dev)
        accumulate = False
    if want_bias and dbias is None:
        dbias = torch.empty(fo, dtype=_f32, device=dev)
    if (defer is not None and row_scale is None and dout is not None and fo > 1 and
            defer.try_add(dout, gate, x, d_n, dw, dbias if want_bias else None, accumulate)):
        return dw, dbias
    ws = _ws(lib().grapes_linear_bwd_weight_gated_workspace_bytes(n, fi, fo), dev)
    _lib.check(lib().grapes_linear_bwd_weight_gated(_p(dout), _p(gate), _p(x), _p(dw), _p(dbias) if want_bias else None, n,
                                                    _p(d_n), fi, fo, 1 if accumulate else 0, _p(row_scale), _p(col_vec),
                                                    _p(dw_head), _p(ws), _stream()),
               "linear_bwd_weight_gated")
    return dw, dbias


def linear_bwd_weight_gated_multi(gates, xs, row_scales, d_ns, col_vec, dw, dbias=None, dw_head=None, accumulate=False):
    """The rank-1 gated dW of SEVERAL hops that share the weights in one split-K GEMM + one slab reduction:
    dw (+)= Σ_h ((row_scales[h] ⊗ col_vec) ⊙ [gates[h] > 0])ᵀ xs[h], dbias (+)= its column sums,
    dw_head (+)= Σ_h row_scales[h]ᵀ gates[h]."""
    import ctypes as C
    nseg = len(gates)
    if not (1 <= nseg <= 4 and len(xs) == nseg and len(row_scales) == nseg and len(d_ns) == nseg):
        raise ValueError("1..4 hops with matching operand lists")
    for t in list(gates) + list(xs) + list(row_scales) + [col_vec, dw]:
        _chk(t, _f32, "operand")
    fo, fi = gates[0].shape[1], xs[0].shape[1]
    arr = lambda ts: (C.c_void_p * nseg)(*[t.data_ptr() for t in ts])
    caps = (C.c_int32 * nseg)(*[x.shape[0] for x in xs])
    ws = _ws(lib().grapes_linear_bwd_weight_gated_workspace_bytes(1, fi, fo), dw.device)
    _lib.check(lib().grapes_linear_bwd_weight_gated_multi(nseg, arr(gates), arr(xs), arr(row_scales), arr(d_ns), caps,
                                                          _p(col_vec), _p(dw), _p(dbias), _p(dw_head), fi, fo,
                                                          1 if accumulate else 0, _p(ws), _stream()),
               "linear_bwd_weight_gated_multi")
    return dw


class DeferredSlabs:
    """Few-row weight gradients of several layers, summed by ONE launch: linear_bwd_weight / linear_bwd_weight_gated called
    with defer=<this> only launch their partial products (when the shape is one of the few-row kernel; otherwise they run
    as usual) and flush() sums all of them.  All deferred calls must be over the same row count (n, d_n)."""

    def __init__(self):
        self.sets, self.n, self.d_n, self.acc = [], None, None, None

    def try_add(self, dout, gate, x, d_n, dw, dbias, accumulate, w=None, dx=None):
        """with (w, dx): the layer's input gradient dx = dout w is computed by the same launch (two independent few-row GEMMs
        side by side) — or nothing is launched and False is returned when either shape is outside its kernel"""
        n, fi = x.shape
        fo = dout.shape[1]
        if len(self.sets) + (2 if dbias is not None else 1) > 8:
            self.flush()
        if self.sets and (self.n != n or (self.d_n is None) != (d_n is None) or
                          (d_n is not None and self.d_n.data_ptr() != d_n.data_ptr()) or self.acc != bool(accumulate)):
            self.flush()
        ws = _ws(lib().grapes_linear_bwd_weight_slabs_bytes(n, fi, fo), x.device)
        if dx is not None:
            rc = lib().grapes_linear_bwd_weight_slabs_and_input(_p(dout), _p(gate), _p(x), _p(w), _p(dx), n, _p(d_n), fi, fo,
                                                                1 if dbias is not None else 0, _p(ws), _stream())
        else:
            rc = lib().grapes_linear_bwd_weight_slabs(_p(dout), _p(gate), _p(x), n, _p(d_n), fi, fo, 1 if dbias is not None else 0,
                                                      _p(ws), _stream())
        if rc != 0:
            return False
        ns = (n + 127) // 128
        self.n, self.d_n, self.acc = n, d_n, bool(accumulate)
        self.sets.append((ws, ws.data_ptr(), dw, fo * fi))
        if dbias is not None:
            self.sets.append((ws, ws.data_ptr() + ns * fo * fi * 4, dbias, fo))
        return True

    def flush(self):
        if not self.sets:
            return
        import ctypes as C
        k = len(self.sets)
        slabs = (C.c_void_p * k)(*[s[1] for s in self.sets])
        outs = (C.c_void_p * k)(*[s[2].data_ptr() for s in self.sets])
        counts = (C.c_int64 * k)(*[s[3] for s in self.sets])
        _lib.check(lib().grapes_slab_reduce_sets(k, slabs, outs, counts, self.n, _p(self.d_n), 1 if self.acc else 0, _stream()),
                   "slab_reduce_sets")
        self.sets = []


class GateBits:
    """What linear_relu_head_fwd_bits keeps of a hidden layer for the backward pass: one word of ReLU gate bits per row and
    32 output columns (include/grapes_hip.h) instead of the [n, f_out] activations."""
    __slots__ = ("words", "n", "f_out")

    def __init__(self, words, n, f_out):
        self.words, self.n, self.f_out = words, n, f_out


def linear_relu_head_fwd_bits(x, w, bias, head_w, d_n=None):
    """head = relu(x wᵀ + bias) head_wᵀ  [n, 1]  WITHOUT writing the activations: -> (GateBits, head).  x may be a
    leading-columns view of a wider matrix.  Only where split_gemm_available(n, f_in, f_out)."""
    _chk(w, _f32, "w"); _chk(bias, _f32, "bias", True); _chk(head_w, _f32, "head_w")
    n, fi = x.shape
    fo = w.shape[0]
    ldx = _row_strided(x, fi, "linear_relu_head_fwd_bits")
    words = torch.empty((n, (fo + 31) // 32), dtype=torch.int32, device=x.device)
    head = torch.empty((n, 1), dtype=_f32, device=x.device)
    _lib.check(lib().grapes_linear_relu_head_fwd_bits(x.data_ptr(), ldx, _p(w), _p(bias), _p(head_w), _p(words), _p(head), n,
                                                      _p(d_n), fi, fo, _stream()), "linear_relu_head_fwd_bits")
    return GateBits(words, n, fo), head


def _dw_cols(dw, fo, fi, what):
    """dw is the padded [f_out, f_in] buffer or the parameter's own [f_out, K] gradient, f_in = K rounded up to a multiple of 4"""
    if dw.dim() != 2 or dw.shape[0] != fo or not dw.is_contiguous() or not (fi - 3 <= dw.shape[1] <= fi):
        raise ValueError(f"{what}: dw must be a dense [f_out, f_in] matrix (or [f_out, K], f_in = K rounded up to a multiple of 4)")
    return int(dw.shape[1])


def linear_bwd_weight_bits_multi(bits, xs, row_scales, d_ns, col_vec, w1, b1, dw, dbias=None, dw_head=None, accumulate=False):
    """Backward of linear_relu_head_fwd_bits for 1..4 row sets that share the weights, given d head = row_scales[h] and the
    head's weight col_vec:  dw (+)= dW1, dbias (+)= db1, dw_head (+)= dW2 (include/grapes_hip.h).  One split-K GEMM + one
    slab reduction; no activation is read."""
    import ctypes as C
    nseg = len(bits)
    if not (1 <= nseg <= 4 and len(xs) == nseg and len(row_scales) == nseg and len(d_ns) == nseg):
        raise ValueError("1..4 row sets with matching operand lists")
    for t in list(row_scales) + [col_vec, dw, w1, b1]:
        _chk(t, _f32, "operand")
    fo, fi = bits[0].f_out, xs[0].shape[1]
    if tuple(w1.shape) != (fo, fi) or not w1.is_contiguous() or b1.numel() != fo:
        raise ValueError("linear_bwd_weight_bits_multi: w1 must be a dense [f_out, f_in] matrix and b1 [f_out]")
    dw_cols = _dw_cols(dw, fo, fi, "linear_bwd_weight_bits_multi")
    strides = (C.c_int32 * nseg)(*[_row_strided(x, fi, "linear_bwd_weight_bits_multi") for x in xs])
    arr = lambda ts: (C.c_void_p * nseg)(*[t.data_ptr() for t in ts])
    caps = (C.c_int32 * nseg)(*[x.shape[0] for x in xs])
    ws = _ws(lib().grapes_linear_bwd_weight_gated_workspace_bytes(1, fi, fo), dw.device)
    _lib.check(lib().grapes_linear_bwd_weight_bits_multi_cols(nseg, arr([b.words for b in bits]), arr(xs), strides, arr(row_scales),
                                                              arr(d_ns), caps, _p(col_vec), _p(w1), _p(b1), _p(dw), dw_cols,
                                                              _p(dbias), _p(dw_head), fi, fo, 1 if accumulate else 0, _p(ws),
                                                              _stream()),
               "linear_bwd_weight_bits_multi")
    return dw


def linear_bwd_weight_bits_pair(bits, xs, row_scales, d_ns, col_vec, w1, b1, dw, dbias, dw_head,
                                bits_b, x_b, row_scale_b, d_n_b, col_vec_b, w1_b, b1_b, dw_b, dbias_b, dw_head_b, accumulate=False):
    """linear_bwd_weight_bits_multi of layer a's 1..3 row sets AND of one row set of a second layer b (same f_out, f_in_b <= f_in_a,
    its own parameters and gradient buffers) in one GEMM launch + one slab reduction."""
    import ctypes as C
    nseg = len(bits)
    if not (1 <= nseg <= 3 and len(xs) == nseg and len(row_scales) == nseg and len(d_ns) == nseg):
        raise ValueError("1..3 row sets of layer a with matching operand lists")
    for t in list(row_scales) + [row_scale_b, col_vec, col_vec_b, dw, dw_b, w1, b1, w1_b, b1_b]:
        _chk(t, _f32, "operand")
    fo, fi, fib = bits[0].f_out, xs[0].shape[1], x_b.shape[1]
    if (bits_b.f_out != fo or tuple(w1.shape) != (fo, fi) or tuple(w1_b.shape) != (fo, fib) or not w1.is_contiguous() or
            not w1_b.is_contiguous()):
        raise ValueError("linear_bwd_weight_bits_pair: dense [f_out, f_in] weights of one f_out")
    allx = list(xs) + [x_b]
    strides = (C.c_int32 * (nseg + 1))(*[_row_strided(x, fi if i < nseg else fib, "linear_bwd_weight_bits_pair") for i, x in enumerate(allx)])
    arr = lambda ts: (C.c_void_p * (nseg + 1))(*[t.data_ptr() for t in ts])
    caps = (C.c_int32 * (nseg + 1))(*[x.shape[0] for x in allx])
    ws = _ws(lib().grapes_linear_bwd_weight_gated_workspace_bytes(1, fi, fo), dw.device)
    dw_cols = _dw_cols(dw, fo, fi, "linear_bwd_weight_bits_pair")
    _lib.check(lib().grapes_linear_bwd_weight_bits_pair_cols(
        nseg, arr([b.words for b in bits] + [bits_b.words]), arr(allx), strides, arr(list(row_scales) + [row_scale_b]),
        arr(list(d_ns) + [d_n_b]), caps, _p(col_vec), _p(w1), _p(b1), _p(dw), dw_cols, _p(dbias), _p(dw_head), fi,
        _p(col_vec_b), _p(w1_b), _p(b1_b), _p(dw_b), _p(dbias_b), _p(dw_head_b), fib, fo, 1 if accumulate else 0, _p(ws),
        _stream()), "linear_bwd_weight_bits_pair")
    return dw, dw_b


def pad_features(X):
    """The resident feature matrix with rows padded to a multiple of 4 floats (16-byte aligned rows for the dwordx4 gathers):
    X itself when its width already is one, else a zero-padded copy (made once, outside the step).  Returns (Xp, F)."""
    F = X.shape[1]
    if F % 4 == 0:
        return X.contiguous(), F
    Xp = torch.zeros((X.shape[0], (F + 3) // 4 * 4), dtype=X.dtype, device=X.device)
    Xp[:, :F] = X
    return Xp, F


class FeaturePlanes:
    """The resident (row-padded) feature matrix split ONCE into its three bf16 planes and registered with the library: the
    gathered-operand bf16x3 GEMMs (linear_fwd_gathered(w_image=...), linear_bwd_weight_gathered(split=True)) then read the planes of
    the rows they gather instead of splitting them in their K loops (X is a run-long constant: main.py:66).  +1.5 x the bytes of X.
    MEASURED SLOWER (Reddit 1.418 against 1.366 ms/step: the kernels are bound by their MFMAs, and the planes are 1.5 x the bytes to
    gather in 8-byte pieces): diagnostic build only, off by default (GRAPES_FEATURE_PLANES=1).
    Keep the object alive as long as X is used; close() (or deletion) forgets the registration."""

    def __init__(self, Xp: torch.Tensor):
        _chk(Xp, _f32, "X")
        if lib().grapes_build_flavor() != b"diag":
            raise _lib.GrapesHipError("FeaturePlanes is an A/B form of the diagnostic build (GRAPES_DIAG=1): it measured slower")
        if Xp.shape[1] % 4 != 0 or Xp.stride(0) != Xp.shape[1]:
            raise ValueError("FeaturePlanes: rows padded to a multiple of 4 floats (ops.pad_features)")
        self.X = Xp
        n, ld = Xp.shape
        self.planes = torch.empty(int(lib().grapes_feature_planes_bytes(n, ld)), dtype=torch.uint8, device=Xp.device)
        _lib.check(lib().grapes_feature_split_planes(_p(Xp), n, ld, _p(self.planes), _stream()), "feature_split_planes")
        _lib.check(lib().grapes_feature_planes_register(_p(Xp), ld, _p(self.planes)), "feature_planes_register")
        self._registered = True

    def close(self):
        if getattr(self, "_registered", False):
            lib().grapes_feature_planes_register(_p(self.X), self.X.shape[1], None)
            self._registered = False

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001  (interpreter shutdown)
            pass


def gcn_aggregate_gather(X, ids, prep, ind_code=None, epoch=0, num_ind=0, d_epoch=None, out=None, F=None):
    """Â · [X[ids] | indicators(ids) | 0-padding] without materialising the gathered features: [n, ceil4(F + num_ind)].
    F: logical feature width when X is a padded matrix (pad_features); default X.shape[1]."""
    peers = X if hasattr(X, "c_table") else None          # peer.PeerFeatures: rows read in place from the GPUs that own them
    if peers is not None:
        _chk(ids, _i32, "ids"); _chk(ind_code, _i32, "ind_code", True)
        n, ldx, F = ids.numel(), peers.pitch, peers.F
        kp = (F + num_ind + 3) // 4 * 4
        if out is None:
            out = torch.empty((n, kp), dtype=_f32, device=ids.device)
        if prep.row_head is None or prep.head_ids is not ids:
            raise ValueError("gcn_aggregate_gather over peer shards needs the graph's head records built on these ids")
        bases, bounds, P = peers.c_table()
        _lib.check(lib().grapes_gcn_aggregate_gather_fwd_peers(bases, bounds, P, F, ldx, _p(ids), _p(ind_code), epoch, _p(d_epoch),
                                                               num_ind, _p(prep.rowptr_t), _p(prep.csr_src), _p(prep.dinv),
                                                               _p(prep.row_head), _p(out), n, _p(prep.d_n), _stream()),
                   "gcn_aggregate_gather_fwd_peers")
        return out
    _chk(X, _f32, "X"); _chk(ids, _i32, "ids"); _chk(ind_code, _i32, "ind_code", True)
    n, ldx = ids.numel(), X.shape[1]
    F = ldx if F is None else int(F)
    if ldx % 4 != 0 or not (ldx - 3 <= F <= ldx):
        raise ValueError("gcn_aggregate_gather: X rows must be padded to a multiple of 4 floats (ops.pad_features)")
    kp = (F + num_ind + 3) // 4 * 4
    if out is None:
        out = torch.empty((n, kp), dtype=_f32, device=X.device)
    head = prep.row_head if (prep.row_head is not None and prep.head_ids is ids) else None     # heads hold THESE ids
    _lib.check(lib().grapes_gcn_aggregate_gather_fwd(_p(X), F, ldx, _p(ids), _p(ind_code), epoch, _p(d_epoch), num_ind,
                                                     _p(prep.rowptr_t), _p(prep.csr_src), _p(prep.dinv), _p(head), _p(out),
                                                     n, _p(prep.d_n), _stream()), "gcn_aggregate_gather_fwd")
    return out


def split_gathered_available(f_out) -> bool:
    return bool(lib().grapes_split_gathered_available(int(f_out)))


def weight_split_image(w, image=None, w_pad=None):
    """bf16x3 image of a weight [f_out, K] (any row stride) for linear_fwd_gathered(w_image=...): one launch per step.
    w_pad [f_out, Kp >= K]: also refreshed as the zero-padded fp32 copy of w by the same launch."""
    _chk(w, _f32, "w") if w.is_contiguous() else None
    fo, k = w.shape
    nbytes = int(lib().grapes_weight_split_image_bytes(k))
    if image is None:
        image = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    if w_pad is not None:
        _chk(w_pad, _f32, "w_pad")
        _lib.check(lib().grapes_weight_split_image_padded(w.data_ptr(), int(w.stride(0)), fo, k, image.data_ptr(), _p(w_pad),
                                                          int(w_pad.shape[1]), _stream()), "weight_split_image_padded")
        return image
    _lib.check(lib().grapes_weight_split_image(w.data_ptr(), int(w.stride(0)), fo, k, image.data_ptr(), _stream()), "weight_split_image")
    return image


def weight_split_images(ws, images, w_pads=None):
    """weight_split_image of up to four weights in ONE launch (images preallocated; w_pads: per weight a padded copy or None)."""
    import ctypes as C
    k = len(ws)
    if not (1 <= k <= 4) or len(images) != k or (w_pads is not None and len(w_pads) != k):
        raise ValueError("weight_split_images: 1..4 weights, as many images")
    pads = list(w_pads) if w_pads is not None else [None] * k
    for w, im, wp in zip(ws, images, pads):
        if w.dtype != _f32 or not w.is_cuda or w.dim() != 2 or w.stride(1) != 1:
            raise _lib.GrapesHipError("weight_split_images: weights must be cuda float32 [f_out, K] with unit column stride")
        if im.numel() * im.element_size() < int(lib().grapes_weight_split_image_bytes(int(w.shape[1]))):
            raise ValueError("weight_split_images: image too small")
        _chk(wp, _f32, "w_pad", True)
    i32s = lambda vs: (C.c_int32 * k)(*[int(v) for v in vs])
    ptrs = lambda ts: (C.c_void_p * k)(*[None if t is None else t.data_ptr() for t in ts])
    _lib.check(lib().grapes_weight_split_images(
        k, ptrs(ws), i32s([w.stride(0) for w in ws]), i32s([w.shape[0] for w in ws]), i32s([w.shape[1] for w in ws]),
        ptrs(images), ptrs(pads), i32s([0 if wp is None else wp.shape[1] for wp in pads]), _stream()), "weight_split_images")
    return images


def linear_fwd_gathered(X, F, ids, w_pad, ind_code=None, epoch=0, num_ind=0, d_epoch=None, d_n=None, out=None, w_image=None):
    """H = [X[ids, :F] | indicators(ids) | 0] · w_padᵀ  — the XW step of a first layer in the reference order, reading the
    frontier rows through the id list.  X: row-padded resident matrix (pad_features); w_pad [f_out, ceil4(F + num_ind)]."""
    _chk(X, _f32, "X"); _chk(ids, _i32, "ids"); _chk(w_pad, _f32, "w_pad"); _chk(ind_code, _i32, "ind_code", True)
    n, ldx, fo = ids.numel(), X.shape[1], w_pad.shape[0]
    kp = (F + num_ind + 3) // 4 * 4
    if w_pad.shape[1] != kp:
        raise ValueError(f"linear_fwd_gathered: weight image must be [f_out, {kp}]")
    if out is None:
        out = torch.empty((n, fo), dtype=_f32, device=X.device)
    if w_image is not None and n >= 8192 and fo % 4 == 0 and _sw("GRAPES_TSPLIT_FWD_TAIL", "1") != "0":
        # bf16x3 on the bf16 matrix pipe, the tiles of the last (partial) round cut along K (A/B: GRAPES_TSPLIT_FWD_TAIL=0)
        ws = _ws(lib().grapes_linear_fwd_gathered_split_tail_workspace_bytes(fo), X.device)
        import ctypes as C
        one = lambda v: (C.c_void_p * 1)(v)
        _lib.check(lib().grapes_linear_fwd_gathered_split_tail(_p(X), F, ldx, _p(ids), 1, one(_p(ind_code) if num_ind else None), epoch,
                                                               _p(d_epoch), (C.c_int32 * 1)(num_ind), one(w_image.data_ptr()),
                                                               one(out.data_ptr()), n, _p(d_n), fo, _p(ws), _stream()),
                   "linear_fwd_gathered_split_tail")
        return out
    if w_image is not None and n >= 8192:   # bf16x3 on the bf16 matrix pipe (w_image = weight_split_image of the same weight);
        # with few rows (the classifier's <= B + hops K, a small graph) a handful of 128-row tiles would walk K alone: the
        # fp32 kernel's split-K form is faster there (Cora, 2.7k rows x K = 1436: 56 us against 107)
        _lib.check(lib().grapes_linear_fwd_gathered_split(_p(X), F, ldx, _p(ids), _p(ind_code), epoch, _p(d_epoch), num_ind,
                                                          w_image.data_ptr(), _p(out), n, _p(d_n), fo, _stream()),
                   "linear_fwd_gathered_split")
        return out
    if w_image is not None and n >= 128 and _sw("GRAPES_TSPLIT_FWD_SPLITK", "1") != "0":
        # few rows: the same bf16x3 kernel split along K into slabs + their sum (instead of the fp32-MFMA split-K kernel)
        ws = _ws(lib().grapes_linear_fwd_gathered_split_k_workspace_bytes(n, kp, fo), X.device)
        _lib.check(lib().grapes_linear_fwd_gathered_split_k(_p(X), F, ldx, _p(ids), _p(ind_code), epoch, _p(d_epoch), num_ind,
                                                            w_image.data_ptr(), _p(out), n, _p(d_n), fo, _p(ws), _stream()),
                   "linear_fwd_gathered_split_k")
        return out
    ws = _ws(lib().grapes_linear_gathered_workspace_bytes(n, kp, fo), X.device)
    _lib.check(lib().grapes_linear_fwd_gathered(_p(X), F, ldx, _p(ids), _p(ind_code), epoch, _p(d_epoch), num_ind, _p(w_pad),
                                                _p(out), n, _p(d_n), fo, _p(ws), _stream()), "linear_fwd_gathered")
    return out


def linear_fwd_gathered_tail(X, F, ids, w_images, f_out, ind_codes, num_inds, epoch=0, d_epoch=None, d_n=None):
    """H_q = [X[ids, :F] | indicators_q(ids) | 0] · W_qᵀ for one or two nets over the SAME gathered rows, on the bf16 matrix pipe
    with a split tail (include/grapes_hip.h: grapes_linear_fwd_gathered_split_tail).  w_images: weight_split_image of each net's
    [f_out, F + num_ind_q] weight.  Returns the list of outputs [n, f_out]."""
    import ctypes as C
    k = len(w_images)
    _chk(X, _f32, "X"); _chk(ids, _i32, "ids")
    for c in ind_codes:
        _chk(c, _i32, "ind_code", True)
    n, ldx = ids.numel(), X.shape[1]
    outs = [torch.empty((n, f_out), dtype=_f32, device=X.device) for _ in range(k)]
    ws = _ws(lib().grapes_linear_fwd_gathered_split_tail_workspace_bytes(f_out), X.device)
    arr = lambda ts: (C.c_void_p * k)(*[None if t is None else t.data_ptr() for t in ts])
    _lib.check(lib().grapes_linear_fwd_gathered_split_tail(_p(X), F, ldx, _p(ids), k, arr(ind_codes), epoch, _p(d_epoch),
                                                           (C.c_int32 * k)(*[int(v) for v in num_inds]), arr(w_images), arr(outs), n,
                                                           _p(d_n), f_out, _p(ws), _stream()), "linear_fwd_gathered_split_tail")
    return outs


def linear_bwd_weight_gathered_multi_ok(F, probs) -> bool:
    """True when linear_bwd_weight_gathered_multi takes these problems (dicts as below): 2..4 of them on the bf16 pipe, f_out and
    padded widths of the tile shape that launch has (include/grapes_hip.h: grapes_linear_bwd_weight_gathered_split_multi)."""
    if not 2 <= len(probs) <= 4:
        return False
    fo = probs[0]["dh"].shape[1]
    for q in probs:
        kp = (F + q.get("num_ind", 0) + 3) // 4 * 4
        if q["dh"].shape[1] != fo or q["dh"].shape[0] != q["ids"].numel() or not q.get("split", True):
            return False
        if tuple(q["dw"].shape) not in ((fo, kp), (fo, F + q.get("num_ind", 0))):
            return False
        if not lib().grapes_linear_bwd_weight_gathered_split_multi_available(fo, kp):
            return False
    return True


def linear_bwd_weight_gathered_multi(X, F, probs, epoch=0, d_epoch=None):
    """Several  dw_q (+)= dh_qᵀ · [X[ids_q, :F] | indicators_q(ids_q) | 0]  in ONE launch and one slab sum (the sampler net's first
    layer at every hop and the log-Z net's at hop 0: main.py:271-283's backward).  probs: dicts with dh, ids, dw and optionally
    ind_code, num_ind, d_n, accumulate, ind_mask — the arguments of linear_bwd_weight_gathered(split=True); problems that name the
    same dw tensor are summed into it together (accumulate: the first one's)."""
    import ctypes as C
    k = len(probs)
    _chk(X, _f32, "X")
    for q in probs:
        _chk(q["dh"], _f32, "dh"); _chk(q["ids"], _i32, "ids"); _chk(q["dw"], _f32, "dw"); _chk(q.get("ind_code"), _i32, "ind_code", True)
    fo, ldx = probs[0]["dh"].shape[1], X.shape[1]
    kps = [(F + q.get("num_ind", 0) + 3) // 4 * 4 for q in probs]
    lds = [0 if tuple(q["dw"].shape) == (fo, kp) else F + q.get("num_ind", 0) for q, kp in zip(probs, kps)]
    n_out = len({q["dw"].data_ptr() for q in probs})
    ws = _ws(lib().grapes_linear_bwd_weight_gathered_split_multi_workspace_bytes(n_out, max(kps), fo), X.device)
    ptrs = lambda ts: (C.c_void_p * k)(*[None if t is None else t.data_ptr() for t in ts])
    i32s = lambda vs: (C.c_int32 * k)(*[int(v) for v in vs])
    _lib.check(lib().grapes_linear_bwd_weight_gathered_split_multi(
        k, ptrs([q["dh"] for q in probs]), _p(X), F, ldx, ptrs([q["ids"] for q in probs]),
        ptrs([q.get("ind_code") if q.get("num_ind", 0) else None for q in probs]), epoch, _p(d_epoch),
        i32s([q.get("num_ind", 0) for q in probs]), (C.c_uint32 * k)(*[int(q.get("ind_mask", 0)) for q in probs]),
        ptrs([q["dw"] for q in probs]), i32s(lds), i32s([q["ids"].numel() for q in probs]), ptrs([q.get("d_n") for q in probs]),
        fo, i32s([1 if q.get("accumulate") else 0 for q in probs]), _p(ws), _stream()), "linear_bwd_weight_gathered_split_multi")


def linear_bwd_weight_gathered(dh, X, F, ids, dw_pad, ind_code=None, epoch=0, num_ind=0, d_epoch=None, d_n=None,
                               accumulate=False, ind_mask=0, split=False):
    """dw_pad [f_out, ceil4(F + num_ind)] (+)= dhᵀ · [X[ids, :F] | indicators(ids) | 0].  ind_mask: the indicator bits the
    forward pass of this layer saw (0 = all) — later hops of the same batch add bits to the shared table."""
    _chk(dh, _f32, "dh"); _chk(X, _f32, "X"); _chk(ids, _i32, "ids"); _chk(dw_pad, _f32, "dw_pad")
    _chk(ind_code, _i32, "ind_code", True)
    n, ldx, fo = ids.numel(), X.shape[1], dh.shape[1]
    kp = (F + num_ind + 3) // 4 * 4
    if split and tuple(dw_pad.shape) == (fo, F + num_ind) and dh.shape[0] == n and F + num_ind != kp:
        # the parameter's own [f_out, F + num_ind] gradient: the slab sum writes it directly (no padded buffer + strided copy)
        ws = _ws(lib().grapes_linear_bwd_weight_gathered_split_workspace_bytes(kp, fo), X.device)
        _lib.check(lib().grapes_linear_bwd_weight_gathered_split_ld(_p(dh), _p(X), F, ldx, _p(ids), _p(ind_code), epoch, _p(d_epoch),
                                                                    num_ind, int(ind_mask), _p(dw_pad), F + num_ind, n, _p(d_n), fo,
                                                                    1 if accumulate else 0, _p(ws), _stream()),
                   "linear_bwd_weight_gathered_split_ld")
        return dw_pad
    if tuple(dw_pad.shape) != (fo, kp) or dh.shape[0] != n:
        raise ValueError("linear_bwd_weight_gathered: shape mismatch")
    if split:                          # bf16x3 on the bf16 matrix pipe
        ws = _ws(lib().grapes_linear_bwd_weight_gathered_split_workspace_bytes(kp, fo), X.device)
        _lib.check(lib().grapes_linear_bwd_weight_gathered_split(_p(dh), _p(X), F, ldx, _p(ids), _p(ind_code), epoch, _p(d_epoch),
                                                                 num_ind, int(ind_mask), _p(dw_pad), n, _p(d_n), fo,
                                                                 1 if accumulate else 0, _p(ws), _stream()),
                   "linear_bwd_weight_gathered_split")
        return dw_pad
    ws = _ws(lib().grapes_linear_gathered_workspace_bytes(n, kp, fo), X.device)
    _lib.check(lib().grapes_linear_bwd_weight_gathered(_p(dh), _p(X), F, ldx, _p(ids), _p(ind_code), epoch, _p(d_epoch), num_ind,
                                                       int(ind_mask), _p(dw_pad), n, _p(d_n), fo, 1 if accumulate else 0, _p(ws), _stream()),
               "linear_bwd_weight_gathered")
    return dw_pad


def linear_bwd_input(dh, w, d_n=None, out=None):
    _chk(dh, _f32, "dh"); _chk(w, _f32, "w")
    n, fo = dh.shape
    fi = w.shape[1]
    if out is None:
        out = torch.empty((n, fi), dtype=_f32, device=dh.device)
    _lib.check(lib().grapes_linear_bwd_input(_p(dh), _p(w), _p(out), n, _p(d_n), fi, fo, _stream()), "linear_bwd_input")
    return out


def _rec_form(prep, n, f) -> bool:
    return (getattr(prep, "head_local", False) and prep.row_head is not None and not prep.items_fwd and 16 < f <= 256 and f % 4 == 0
            and prep.row_head.shape[0] >= n and _sw("GRAPES_AGG_REC", "1") != "0")


def gcn_aggregate_fwd(h, prep: PreparedGraph, bias=None, relu=False, out=None):
    _chk(h, _f32, "h"); _chk(bias, _f32, "bias", True)
    n, f = h.shape
    if out is None:
        out = torch.empty_like(h)
    if _rec_form(prep, n, f) and (bias is None or bias.data_ptr() % 16 == 0):
        # the hop graph carries head records over local ids: one dependent trip per row (grapes_gcn_aggregate_fwd_rec)
        _lib.check(lib().grapes_gcn_aggregate_fwd_rec(_p(h), _p(prep.row_head), _p(prep.rowptr_t), _p(prep.csr_src), _p(prep.dinv),
                                                      _p(bias), _p(out), n, _p(prep.d_n), f, 1 if relu else 0, None, None, None,
                                                      _stream()), "gcn_aggregate_fwd_rec")
        return out
    use_items = prep.items_fwd and f > 16 and prep.n > _SMALL_GRAPH
    ws = _ws(lib().grapes_gcn_aggregate_workspace_bytes(prep.item_cap, f), h.device) if use_items else None
    _lib.check(lib().grapes_gcn_aggregate_fwd(_p(h), _p(prep.rowptr_t), _p(prep.csr_src), _p(prep.dinv), _p(bias), _p(out),
                                              n, _p(prep.d_n), f, 1 if relu else 0,
                                              _p(prep.items_t) if use_items else None,
                                              _p(prep.n_items_t) if use_items else None,
                                              prep.item_cap if use_items else 0, _p(ws), _stream()), "gcn_aggregate_fwd")
    return out


def gcn_aggregate_fwd_head(h, prep: PreparedGraph, bias, relu, head_w, want_bits=False):
    """(Â h + bias [ReLU], its product with head_w [f][, gate bits]) in one launch — the aggregation of a transform-first layer
    and the X W step of the 1-wide layer behind it (main.py:210).  want_bits (ReLU, f <= 256): also int32 [n, 8], the ReLU gates
    of the output for gcn_aggregate_bwd_rank1(gate_bits=...).  None when the shape is not covered (the caller runs the two
    launches)."""
    _chk(h, _f32, "h"); _chk(bias, _f32, "bias", True); _chk(head_w, _f32, "head_w")
    n, f = h.shape
    if f <= 16 or f % 4 or head_w.numel() != f or (prep.items_fwd and prep.n > _SMALL_GRAPH):
        return None
    out = torch.empty_like(h)
    hw = torch.empty((n, 1), dtype=_f32, device=h.device)
    bits = torch.empty((n, 8), dtype=_i32, device=h.device) if (want_bits and relu and f <= 256) else None
    if _rec_form(prep, n, f) and (bias is None or bias.data_ptr() % 16 == 0):
        _lib.check(lib().grapes_gcn_aggregate_fwd_rec(_p(h), _p(prep.row_head), _p(prep.rowptr_t), _p(prep.csr_src), _p(prep.dinv),
                                                      _p(bias), _p(out), n, _p(prep.d_n), f, 1 if relu else 0, _p(head_w), _p(hw),
                                                      _p(bits), _stream()), "gcn_aggregate_fwd_rec")
        return (out, hw, bits) if want_bits else (out, hw)
    _lib.check(lib().grapes_gcn_aggregate_fwd_head(_p(h), _p(prep.rowptr_t), _p(prep.csr_src), _p(prep.dinv), _p(bias), _p(out),
                                                   n, _p(prep.d_n), f, 1 if relu else 0, _p(head_w), _p(hw), _p(bits), _stream()),
               "gcn_aggregate_fwd_head")
    return (out, hw, bits) if want_bits else (out, hw)


def gcn_aggregate_bwd_rank1(act, dh2, w2, prep: PreparedGraph, dw_head=None, dbias=None, accumulate=False, gate_bits=None):
    """Backward of (transform-first GCNConv -> ReLU -> 1-wide GCNConv) from dh2 = Âᵀ d(head output): returns
    dh = Âᵀ ((dh2 ⊗ w2) ⊙ [act > 0]) [n, f]; dw_head (+)= dh2ᵀ act, dbias (+)= the first layer's bias gradient.  No n x f
    temporary is written (include/grapes_hip.h: grapes_gcn_aggregate_bwd_rank1)."""
    _chk(act, _f32, "act"); _chk(dh2, _f32, "dh2"); _chk(w2, _f32, "w2"); _chk(dw_head, _f32, "dw_head", True); _chk(dbias, _f32, "dbias", True)
    n, f = act.shape
    if dh2.numel() != n or w2.numel() != f:
        raise ValueError("gcn_aggregate_bwd_rank1: dh2 [n], w2 [f]")
    dh = torch.empty_like(act)
    use_items = prep.n > _SMALL_GRAPH
    ws = _ws(lib().grapes_gcn_aggregate_bwd_rank1_workspace_bytes(prep.item_cap, f), act.device)
    if gate_bits is not None:     # the gates of the aggregation from 32 bytes of bits per row (gcn_aggregate_fwd_head)
        if gate_bits.dtype != _i32 or tuple(gate_bits.shape) != (n, 8) or not gate_bits.is_contiguous() or f > 256:
            raise ValueError("gcn_aggregate_bwd_rank1: gate_bits int32 [n, 8], f <= 256")
        _lib.check(lib().grapes_gcn_aggregate_bwd_rank1_bits(
            _p(act), _p(gate_bits), _p(dh2), _p(w2), _p(prep.rowptr_s), _p(prep.csr_dst), _p(prep.dinv), _p(dh), _p(dw_head),
            _p(dbias), 1 if accumulate else 0, n, _p(prep.d_n), f, _p(prep.items_s) if use_items else None,
            _p(prep.n_items_s) if use_items else None, prep.item_cap if use_items else 0, _p(ws), _stream()),
            "gcn_aggregate_bwd_rank1_bits")
        return dh
    _lib.check(lib().grapes_gcn_aggregate_bwd_rank1(_p(act), _p(dh2), _p(w2), _p(prep.rowptr_s), _p(prep.csr_dst), _p(prep.dinv),
                                                    _p(dh), _p(dw_head), _p(dbias), 1 if accumulate else 0, n, _p(prep.d_n), f,
                                                    _p(prep.items_s) if use_items else None,
                                                    _p(prep.n_items_s) if use_items else None,
                                                    prep.item_cap if use_items else 0, _p(ws), _stream()),
               "gcn_aggregate_bwd_rank1")
    return dh


def gcn_aggregate_bwd_rank1_multi(problems):
    """gcn_aggregate_bwd_rank1 (gate-bit form) for up to three independent problems in the same five launches.  problems: dicts
    with act, dh2, w2, prep, gate_bits and optionally dw_head, dbias, accumulate.  Returns the list of dh.  Equal bit for bit to
    the calls one after the other, in order (problems that name the same dw_head / dbias are added in that order)."""
    import ctypes as C
    k = len(problems)
    if not (1 <= k <= 3):
        raise ValueError("gcn_aggregate_bwd_rank1_multi: 1..3 problems")
    f = problems[0]["act"].shape[1]
    dev = problems[0]["act"].device
    dhs, wss, ns, caps, items, nitems, accs = [], [], [], [], [], [], []
    for pr in problems:
        act, dh2, w2, prep, bits = pr["act"], pr["dh2"], pr["w2"], pr["prep"], pr["gate_bits"]
        _chk(act, _f32, "act"); _chk(dh2, _f32, "dh2"); _chk(w2, _f32, "w2")
        _chk(pr.get("dw_head"), _f32, "dw_head", True); _chk(pr.get("dbias"), _f32, "dbias", True)
        n = act.shape[0]
        if act.shape[1] != f or dh2.numel() != n or w2.numel() != f or f > 256:
            raise ValueError("gcn_aggregate_bwd_rank1_multi: one width f <= 256; dh2 [n], w2 [f]")
        if bits is None or bits.dtype != _i32 or tuple(bits.shape) != (n, 8) or not bits.is_contiguous():
            raise ValueError("gcn_aggregate_bwd_rank1_multi: gate_bits int32 [n, 8]")
        use_items = prep.n > _SMALL_GRAPH
        dhs.append(torch.empty_like(act))
        wss.append(_ws(lib().grapes_gcn_aggregate_bwd_rank1_workspace_bytes(prep.item_cap, f), dev))
        ns.append(n); caps.append(prep.item_cap if use_items else 0)
        items.append(prep.items_s if use_items else None); nitems.append(prep.n_items_s if use_items else None)
        accs.append(1 if pr.get("accumulate") else 0)
    arr = lambda ts: (C.c_void_p * k)(*[None if t is None else t.data_ptr() for t in ts])
    ints = lambda vs: (C.c_int32 * k)(*vs)
    _lib.check(lib().grapes_gcn_aggregate_bwd_rank1_bits_multi(
        k, arr([p["act"] for p in problems]), arr([p["gate_bits"] for p in problems]), arr([p["dh2"] for p in problems]),
        arr([p["w2"] for p in problems]), arr([p["prep"].rowptr_s for p in problems]), arr([p["prep"].csr_dst for p in problems]),
        arr([p["prep"].dinv for p in problems]), arr(dhs), arr([p.get("dw_head") for p in problems]),
        arr([p.get("dbias") for p in problems]), ints(accs), ints(ns), arr([p["prep"].d_n for p in problems]), f, arr(items),
        arr(nitems), ints(caps), arr(wss), _stream()), "gcn_aggregate_bwd_rank1_bits_multi")
    return dhs


def scale_rows(h, dinv, out=None):
    """hs[r, :] = dinv[r] * h[r, :] (out may be h): the pre-scaled operand of gcn_aggregate_fwd(..., prescaled=True)."""
    _chk(h, _f32, "h"); _chk(dinv, _f32, "dinv")
    n, f = h.shape
    if out is None:
        out = torch.empty_like(h)
    _lib.check(lib().grapes_scale_rows(_p(h), _p(dinv), _p(out), n, f, _stream()), "scale_rows")
    return out


def gcn_aggregate_fwd_prescaled(hs, prep: PreparedGraph, bias=None, relu=False, out=None):
    """Â h + bias from hs = scale_rows(h, prep.dinv): out[c] = dinv[c] (sum hs[s] + hs[c]) + b — the full-batch inference form
    (no gather of dinv[source] per aggregated edge)."""
    _chk(hs, _f32, "hs"); _chk(bias, _f32, "bias", True)
    n, f = hs.shape
    if out is None:
        out = torch.empty_like(hs)
    use_items = prep.items_fwd and prep.n > _SMALL_GRAPH
    ws = _ws(lib().grapes_gcn_aggregate_workspace_bytes(prep.item_cap, f), hs.device) if use_items else None
    _lib.check(lib().grapes_gcn_aggregate_fwd_prescaled(_p(hs), _p(prep.rowptr_t), _p(prep.csr_src), _p(prep.dinv), _p(bias),
                                                        _p(out), n, _p(prep.d_n), f, 1 if relu else 0,
                                                        _p(prep.items_t) if use_items else None,
                                                        _p(prep.n_items_t) if use_items else None,
                                                        prep.item_cap if use_items else 0, _p(ws), _stream()),
               "gcn_aggregate_fwd_prescaled")
    return out


def gcn_aggregate_narrow_pair(h_a, h_b, prep: PreparedGraph, bias_a=None, bias_b=None):
    """(Â h_a + bias_a, Â h_b + bias_b) for two [n, 1] vectors over the same prepared graph in one launch."""
    _chk(h_a, _f32, "h_a"); _chk(h_b, _f32, "h_b"); _chk(bias_a, _f32, "bias_a", True); _chk(bias_b, _f32, "bias_b", True)
    if h_a.numel() != h_b.numel():
        raise ValueError("gcn_aggregate_narrow_pair: two vectors of one length")
    out_a, out_b = torch.empty_like(h_a), torch.empty_like(h_b)
    _lib.check(lib().grapes_gcn_aggregate_narrow_pair(_p(h_a), _p(h_b), _p(prep.rowptr_t), _p(prep.csr_src), _p(prep.dinv),
                                                      _p(bias_a), _p(bias_b), _p(out_a), _p(out_b), h_a.numel(), _p(prep.d_n),
                                                      _stream()), "gcn_aggregate_narrow_pair")
    return out_a, out_b


_BWD_HOSTS: List = []     # launches waiting to carry a few-row backward aggregation (carry_backward_aggregations)


def carry_backward_aggregations(hosts):
    """hosts: callables, each issuing ONE launch that can carry a recorded few-row backward aggregation as extra workgroups
    (SamplerHeadBwdMulti.launch(1) / (2)) and that neither reads what the next gcn_aggregate_bwd calls write nor writes what
    they read.  Each of the next gcn_aggregate_bwd calls then RECORDS its launch and issues the next host with the record
    attached: the pair takes the time of the longer one (both are dependent round trips on a mostly idle chip).  Order on the
    stream: host launch (with the aggregation inside), then whatever the caller issues next.  flush_backward_hosts() issues the
    hosts nobody used.  Not while another rider program is attached."""
    _BWD_HOSTS[:] = list(hosts)


def flush_backward_hosts():
    while _BWD_HOSTS:
        _BWD_HOSTS.pop(0)()


def gcn_aggregate_bwd(dout, prep: PreparedGraph, relu_out=None, want_bias=True, dbias=None, accumulate_bias=False):
    """Returns (dh, dbias).  dout is not modified."""
    _chk(dout, _f32, "dout"); _chk(relu_out, _f32, "relu_out", True)
    n, f = dout.shape
    dev = dout.device
    dpre = torch.empty_like(dout) if relu_out is not None else dout
    dh = torch.empty_like(dout)
    if want_bias and dbias is None:
        dbias = torch.empty(f, dtype=_f32, device=dev)
        accumulate_bias = False
    ws = _ws(lib().grapes_gcn_aggregate_bwd_workspace_bytes(prep.item_cap, f), dev)
    use_items = prep.n > _SMALL_GRAPH          # small graphs: one launch, every row by one wavefront whatever its length
    L = lib()
    host = _BWD_HOSTS.pop(0) if (_BWD_HOSTS and not use_items) else None
    if host is not None:
        _lib.check(L.grapes_rider_record_begin(), "rider_record_begin")
    prog = -1
    try:
        rc = L.grapes_gcn_aggregate_bwd(_p(dout), _p(relu_out), _p(prep.rowptr_s), _p(prep.csr_dst), _p(prep.dinv),
                                        _p(dpre), _p(dh), _p(dbias) if want_bias else None,
                                        1 if accumulate_bias else 0, n, _p(prep.d_n), f,
                                        _p(prep.items_s) if use_items else None,
                                        _p(prep.n_items_s) if use_items else None,
                                        prep.item_cap if use_items else 0, _p(ws), _p(_ticket(dev)[16:32]), _stream())
    finally:
        if host is not None:
            prog = int(L.grapes_rider_record_end())
    _lib.check(rc, "gcn_aggregate_bwd")
    if host is not None:        # (a call that took another form launched at once and recorded nothing: the host then runs on its own)
        _lib.check(L.grapes_rider_attach(prog, 0, _stream()), "rider_attach")
        try:
            host()
        finally:
            alone = L.grapes_rider_detach(_stream(), None)
            L.grapes_rider_free(prog)
        if alone < 0:
            raise _lib.GrapesHipError(f"rider_detach failed ({alone})")
    return dh, dbias


# ------------------------------------------------------------------------------- sampler
def gumbel_topk(logits, k, uniforms=None, logit_index=None, candidate_ids=None, n=None, d_n=None, mode=0,
                philox_seed=0, philox_offset=0, d_philox_offset=None, want_log_prob=True, want_keys=False,
                want_stats=True, prefix_ids=None, stats_out=None, agg=None, defer_finish=False, ext=None):
    """Sampler draw (two launches).  Returns dict(mask, kept_pos, kept_ids, kept_count, log_prob, keys, stats);
    defer_finish: the draw's last launch has no tail — res["finish"] must be handed to the NEXT frontier_expand_fused(finish=...)
    on the stream, whose extra workgroup forms stats[4] (the log-prob sum) and zeroes the draw-wide histogram; until then stats[4]
    is not valid and no other draw may start on the device.
    with prefix_ids also union_ids = [prefix_ids | kept ids] and union_count (main.py:236-238).
    agg = (head_in [n_rows] or [n_rows, 1], prep, bias [1], cand_pos): `logits` is None and the logits are produced on the way,
    logits[r] = (Â head_in)[r] + bias over the prepared hop graph (the sampler net's 1-wide last layer), fused with the key
    computation; logit_index is then required (nb_local).  The result carries them as res["logits"] ([n_rows, 1])."""
    if agg is not None:
        head_in, prep, bias, cand_pos = agg
        _chk(head_in, _f32, "head_in"); _chk(bias, _f32, "bias", True); _chk(cand_pos, _i32, "cand_pos")
        if logits is not None or logit_index is None:
            raise ValueError("gumbel_topk(agg=...): logits must be None and logit_index given")
        n_rows = head_in.numel()
        logits = torch.empty(n_rows, dtype=_f32, device=head_in.device)
    _chk(logits, _f32, "logits"); _chk(uniforms, _f32, "uniforms", True)
    _chk(logit_index, _i32, "logit_index", True); _chk(candidate_ids, _i32, "candidate_ids", True)
    _chk(prefix_ids, _i32, "prefix_ids", True)
    dev = logits.device
    if n is None:
        n = logit_index.numel() if logit_index is not None else logits.numel()
    if uniforms is not None and uniforms.numel() < n:
        raise ValueError("uniforms shorter than the candidate list")
    if prefix_ids is not None and candidate_ids is None:
        raise ValueError("prefix_ids needs candidate_ids")
    kk = min(k, n) if n > 0 else 0
    mask = torch.empty(n, dtype=_f32, device=dev)
    kept_pos = torch.empty(max(kk, 1), dtype=_i32, device=dev)
    kept_ids = torch.empty(max(kk, 1), dtype=_i32, device=dev) if candidate_ids is not None else None
    cnt = torch.empty(1, dtype=_i32, device=dev)
    log_prob = torch.empty(n, dtype=_f32, device=dev) if want_log_prob else None
    keys = torch.empty(n, dtype=_f32, device=dev) if want_keys else None
    stats = (stats_out if stats_out is not None else torch.empty(6, dtype=_f32, device=dev)) if want_stats else None
    _chk(stats, _f32, "stats", True)
    npre = prefix_ids.numel() if prefix_ids is not None else 0
    union = torch.empty(npre + max(kk, 1), dtype=_i32, device=dev) if prefix_ids is not None else None
    ucnt = torch.empty(1, dtype=_i32, device=dev) if prefix_ids is not None else None
    ws = _ws(lib().grapes_sampler_workspace_bytes(n), dev)
    finish = None
    if agg is not None:
        _lib.check(lib().grapes_gumbel_topk_from_aggregate(
            _p(head_in), _p(prep.rowptr_t), _p(prep.csr_src), _p(prep.dinv), _p(bias), _p(logits), n_rows, _p(prep.d_n),
            _p(cand_pos), _p(logit_index), _p(uniforms), philox_seed, philox_offset, _p(d_philox_offset), n, _p(d_n), k, mode,
            _p(candidate_ids), _p(mask), _p(kept_pos), _p(kept_ids), _p(cnt), _p(log_prob), _p(keys), _p(stats),
            _p(prefix_ids), npre, _p(union), _p(ucnt), _p(ws), _stream()), "gumbel_topk_from_aggregate")
    elif _SAMPLER_GHIST and defer_finish and n > 0:
        import ctypes as C
        fin = _DrawFinishArgs()
        e_rowptr = e_prefix = union_ext = None
        if ext is not None:       # ext = (rowptr of the graph expanded next, prefix ids' extents int64[2 npre]): res["union_ext"]
            e_rowptr, e_prefix = ext
            _chk(e_rowptr, _i64, "ext rowptr"); _chk(e_prefix, _i64, "ext prefix", npre == 0)
            if prefix_ids is None or (npre and e_prefix.numel() < 2 * npre):
                raise ValueError("gumbel_topk(ext=...): needs prefix_ids and two int64 per prefix id")
            union_ext = torch.empty(2 * (npre + max(kk, 1)), dtype=torch.int64, device=dev)
        _lib.check(lib().grapes_gumbel_topk_deferred_ext(_p(logits), _p(logit_index), _p(uniforms), philox_seed, philox_offset,
                                                         _p(d_philox_offset), n, _p(d_n), k, mode, _p(candidate_ids), _p(mask),
                                                         _p(kept_pos), _p(kept_ids), _p(cnt), _p(log_prob), _p(keys), _p(stats),
                                                         _p(prefix_ids), npre, _p(union), _p(ucnt), _p(ws), _p(_sampler_hist(dev)),
                                                         C.byref(fin), _p(e_rowptr), _p(e_prefix), _p(union_ext), _stream()),
                   "gumbel_topk_deferred_ext")
        finish = (fin, ws, stats)             # (the workspace must outlive the launch that finishes the draw)
    elif _SAMPLER_GHIST:
        _lib.check(lib().grapes_gumbel_topk_hist(_p(logits), _p(logit_index), _p(uniforms), philox_seed, philox_offset,
                                                 _p(d_philox_offset), n, _p(d_n), k, mode, _p(candidate_ids), _p(mask),
                                                 _p(kept_pos), _p(kept_ids), _p(cnt), _p(log_prob), _p(keys), _p(stats),
                                                 _p(prefix_ids), npre, _p(union), _p(ucnt), _p(ws), _p(_sampler_hist(dev)), _stream()),
                   "gumbel_topk_hist")
    else:
        _lib.check(lib().grapes_gumbel_topk(_p(logits), _p(logit_index), _p(uniforms), philox_seed, philox_offset,
                                            _p(d_philox_offset), n, _p(d_n), k, mode, _p(candidate_ids), _p(mask),
                                            _p(kept_pos), _p(kept_ids), _p(cnt), _p(log_prob), _p(keys), _p(stats),
                                            _p(prefix_ids), npre, _p(union), _p(ucnt), _p(ws), _stream()), "gumbel_topk")
    out = dict(mask=mask, kept_pos=kept_pos[:kk], kept_ids=None if kept_ids is None else kept_ids[:kk], kept_count=cnt,
               log_prob=log_prob, keys=keys, stats=stats)
    if prefix_ids is not None:
        out["union_ids"], out["union_count"] = union[:npre + kk], ucnt
    if agg is not None:
        out["logits"] = logits.view(-1, 1)
    if finish is not None:
        out["finish"] = finish
        if ext is not None:
            out["union_ext"] = union_ext
    return out


_TICKETS = {}
_SAMPLER_HIST = {}
_SAMPLER_GHIST = _sw("GRAPES_SAMPLER_GHIST", "1") != "0"     # A/B: one 12-bit histogram per draw instead of per-workgroup 8-bit rows


def _sampler_hist(dev) -> torch.Tensor:
    """The per-device, zero-at-rest histogram of grapes_gumbel_topk_hist (draws on one device are stream-ordered)."""
    t = _SAMPLER_HIST.get(dev)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("sampler histogram: first use inside a stream capture; run one eager draw first")
        t = torch.zeros(int(lib().grapes_sampler_hist_words()), dtype=_i32, device=dev)
        _SAMPLER_HIST[dev] = t
    return t


def _ticket(dev) -> torch.Tensor:
    """A persistent zero word per device for the last-workgroup tickets (kernels leave it zero)."""
    t = _TICKETS.get(dev)
    if t is None:
        t = torch.zeros(64, dtype=_i32, device=dev)     # [0] sums, [1] sampler head bwd, [8] step losses, [16:32] column sums
        _TICKETS[dev] = t
    return t


def bernoulli_logprob_bwd(logits, mask, grad_vec=None, d_grad_scale=None, logit_index=None, out=None, d_n=None,
                          sum_out=None, accumulate_sum=False):
    """sum_out: float32[1] that receives (+)= the sum of the gradients written (bias gradient of a 1-wide head)."""
    _chk(logits, _f32, "logits"); _chk(mask, _f32, "mask"); _chk(sum_out, _f32, "sum_out", True)
    n = mask.numel()
    if out is None:
        out = torch.zeros_like(logits) if logit_index is not None else torch.empty_like(logits)
    partials = torch.empty(2048, dtype=_f32, device=logits.device) if sum_out is not None else None
    ticket = _ticket(logits.device) if sum_out is not None else None
    _lib.check(lib().grapes_bernoulli_logprob_bwd(_p(logits), _p(logit_index), _p(mask), _p(grad_vec), _p(d_grad_scale),
                                                  _p(out), n, _p(d_n), _p(sum_out), 1 if accumulate_sum else 0,
                                                  _p(partials), _p(ticket), _stream()), "bernoulli_logprob_bwd")
    return out


class SamplerHeadBwdMulti:
    """Backward of the sampler net's 1-wide head for up to four hops in two launches: per hop the dense d log_prob / d logit
    (zero on non-candidate rows; cand_pos from frontier_compact(want_cand_pos=True)) and its by-source aggregation
    Âᵀ dlogits; sum_out (+)= the sum of all dlogits.  A hop whose mask is None is a MEAN head (the log-Z net): d logits =
    scale / n on its live rows, their sum goes to mean_sum_out.  Results: .dlog, .dh ([count, n_cap] each) once launch(0), or
    launch(1) and launch(2), have been issued — between the two the caller may issue launches of its own, and each of them
    carries a pending recorded few-row backward aggregation (carry_backward_aggregations)."""

    def __init__(self, logits, masks, cand_pos, preps, d_grad_scale=None, sum_out=None, accumulate_sum=False, mean_sum_out=None):
        k = len(logits)
        if not (1 <= k <= 4) or len(masks) != k or len(cand_pos) != k or len(preps) != k:
            raise ValueError("sampler_head_bwd_multi: 1..4 hops")
        dev = logits[0].device
        for l, m, c in zip(logits, masks, cand_pos):
            _chk(l, _f32, "logits"); _chk(m, _f32, "mask", True); _chk(c, _i32, "cand_pos", m is None)
        _chk(mean_sum_out, _f32, "mean_sum_out", True)
        _chk(sum_out, _f32, "sum_out", True); _chk(d_grad_scale, _f32, "d_grad_scale", True)
        self.k, self.dev = k, dev
        self.caps = [int(l.numel()) for l in logits]
        ncap = max(self.caps)
        self.dlog = torch.empty((k, ncap), dtype=_f32, device=dev)
        self.dh = torch.empty((k, ncap), dtype=_f32, device=dev)
        self.ws = _ws(lib().grapes_sampler_head_bwd_multi_workspace_bytes(), dev)
        self._hold = (list(logits), list(masks), list(cand_pos), list(preps), d_grad_scale, sum_out, mean_sum_out)
        self.accumulate_sum = accumulate_sum

    def launch(self, phase: int = 0):
        import ctypes as C
        logits, masks, cand_pos, preps, d_grad_scale, sum_out, mean_sum_out = self._hold
        k, dlog, dh = self.k, self.dlog, self.dh
        arr = lambda ts: (C.c_void_p * k)(*[None if t is None else t.data_ptr() for t in ts])
        _lib.check(lib().grapes_sampler_head_bwd_multi_phase(
            k, arr(logits), arr(masks), arr(cand_pos), (C.c_int32 * k)(*self.caps), arr([p.d_n for p in preps]), _p(d_grad_scale),
            arr([p.rowptr_s for p in preps]), arr([p.csr_dst for p in preps]), arr([p.dinv for p in preps]),
            arr([dlog[q] for q in range(k)]), arr([dh[q] for q in range(k)]), _p(sum_out), 1 if self.accumulate_sum else 0,
            _p(mean_sum_out), _p(self.ws), _p(_ticket(self.dev)[1:2]), int(phase), _stream()), "sampler_head_bwd_multi")


def sampler_head_bwd_multi(logits, masks, cand_pos, preps, d_grad_scale=None, sum_out=None, accumulate_sum=False,
                           mean_sum_out=None):
    """SamplerHeadBwdMulti in one call.  Returns (dlogits [count, n_cap], dh [count, n_cap])."""
    hb = SamplerHeadBwdMulti(logits, masks, cand_pos, preps, d_grad_scale=d_grad_scale, sum_out=sum_out,
                             accumulate_sum=accumulate_sum, mean_sum_out=mean_sum_out)
    hb.launch(0)
    return hb.dlog, hb.dh


def philox_uniform(n, seed, offset, device):
    out = torch.empty(n, dtype=_f32, device=device)
    _lib.check(lib().grapes_philox_uniform(_p(out), n, seed, offset, _stream()), "philox_uniform")
    return out


def fill(x, value=0.0, d_n=None, d_value=None, scale_by_inv_n=0.0, sum_out=None, accumulate_sum=False):
    """x[i] = value (or *d_value, optionally times scale_by_inv_n / n) for i < n; sum_out (+)= n * that value."""
    _chk(x, _f32, "x"); _chk(sum_out, _f32, "sum_out", True)
    _lib.check(lib().grapes_fill(_p(x), x.numel(), _p(d_n), float(value), _p(d_value), float(scale_by_inv_n), _p(sum_out),
                                 1 if accumulate_sum else 0, _stream()), "fill")
    return x


def reduce_sum(x, mean=False, d_n=None):
    _chk(x, _f32, "x")
    out = torch.empty(1, dtype=_f32, device=x.device)
    _lib.check(lib().grapes_reduce_sum(_p(x), x.numel(), _p(d_n), 1 if mean else 0, _p(out), _stream()), "reduce_sum")
    return out


# ------------------------------------------------------------------------------- ingest (§8f N3)
def csr_build(edge_index, num_nodes, status=None):
    """(rowptr int64[N+1], col int32[nnz]) of sp.csr_matrix((ones(E, bool), edge_index), (N, N)) — main.py:134-136 —
    from a device int64 edge_index [2, E].  One host read (nnz) to trim the column array."""
    if not edge_index.is_cuda or edge_index.dtype != _i64:
        raise _lib.GrapesHipError("csr_build: edge_index must be a cuda int64 tensor (torch.long, as the reference's data.edge_index)")
    if edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError("edge_index must be [2, E]")
    dev, E, N = edge_index.device, int(edge_index.shape[1]), int(num_nodes)
    src, dst = edge_index[0].contiguous(), edge_index[1].contiguous()
    rowptr = torch.empty(N + 1, dtype=_i64, device=dev)
    col = torch.empty(max(E, 1), dtype=_i32, device=dev)
    nnz = torch.zeros(1, dtype=_i64, device=dev)
    own = status is None
    if own:
        status = torch.zeros(1, dtype=_i32, device=dev)
    ws = torch.empty(int(lib().grapes_csr_build_workspace_bytes(E, N)) + 256, dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    _lib.check(lib().grapes_csr_build(_p(src), _p(dst), E, N, _p(rowptr), _p(col), _p(nnz), ws.data_ptr() + off, _p(status),
                                      _stream()), "csr_build")
    k = int(nnz.item())
    if own and int(status.item()):
        raise _lib.GrapesHipError("csr_build: edge_index holds node ids outside [0, num_nodes)")
    del ws
    return rowptr, col[:k].clone() if k < col.numel() // 2 else col[:k]


# ------------------------------------------------------------------------------- 1-D partition exchange (§8e)
def exchange_pack_query(ids32, d_n, cap, query):
    """query[cap + 1] <- [ids | padding | live count] in one launch."""
    _chk(ids32, _i32, "ids"); _chk(query, _i32, "query"); _chk(d_n, _i32, "d_n", True)
    if query.numel() != cap + 1:
        raise ValueError("exchange_pack_query: query must hold cap + 1 words")
    _lib.check(lib().grapes_exchange_pack_query(_p(ids32), ids32.numel(), _p(d_n), cap, _p(query), _stream()),
               "exchange_pack_query")


def exchange_serve_rows(rowptr_local, col_local, req, n_peers, cap, lo, hi, reply, reply_stride, e_slot, status=None):
    """Owner side of the adjacency-row exchange: fills the per-peer reply slots [len | off | columns]."""
    _chk(rowptr_local, _i64, "rowptr_local"); _chk(col_local, _i32, "col_local"); _chk(req, _i32, "req")
    _chk(reply, _i32, "reply")
    if req.numel() != n_peers * (cap + 1) or reply.numel() != n_peers * reply_stride:
        raise ValueError("exchange_serve_rows: req / reply do not match (n_peers, cap, reply_stride)")
    eoff = torch.empty(n_peers * cap + 1, dtype=_i32, device=req.device)
    _lib.check(lib().grapes_exchange_serve_rows(_p(rowptr_local), _p(col_local), _p(req), n_peers, cap, lo, hi, _p(reply),
                                                reply_stride, e_slot, _p(eoff), _p(status), _stream()),
               "exchange_serve_rows")


def exchange_recv_rows(back, reply_stride, nodes, bounds, n_peers, e_cap, d_m=None, status=None):
    """Requester side: (src, dst, d_e, eoff) with the contract of frontier_offsets + frontier_expand."""
    _chk(back, _i32, "back"); _chk(nodes, _i32, "nodes"); _chk(bounds, _i32, "bounds")
    cap, dev = nodes.numel(), nodes.device
    if back.numel() != n_peers * reply_stride or bounds.numel() != n_peers + 1:
        raise ValueError("exchange_recv_rows: back / bounds do not match (n_peers, reply_stride)")
    eoff = torch.empty(cap + 1, dtype=_i32, device=dev)
    rowstart = torch.empty(cap, dtype=_i32, device=dev)
    src = torch.empty(e_cap, dtype=_i32, device=dev)
    dst = torch.empty(e_cap, dtype=_i32, device=dev)
    d_e = torch.empty(1, dtype=_i32, device=dev)
    _lib.check(lib().grapes_exchange_recv_rows(_p(back), reply_stride, _p(nodes), cap, _p(d_m), _p(bounds), n_peers, e_cap,
                                               _p(eoff), _p(rowstart), _p(src), _p(dst), _p(d_e), _p(status), _stream()),
               "exchange_recv_rows")
    return src, dst, d_e, eoff


def exchange_serve_features(X_local, req, n_peers, cap, lo, hi, reply, n_slot, status=None):
    _chk(X_local, _f32, "X_local"); _chk(req, _i32, "req"); _chk(reply, _f32, "reply")
    F = X_local.shape[1]
    if req.numel() != n_peers * (cap + 1) or reply.numel() != n_peers * n_slot * F:
        raise ValueError("exchange_serve_features: req / reply do not match (n_peers, cap, n_slot, F)")
    _lib.check(lib().grapes_exchange_serve_features(_p(X_local), F, _p(req), n_peers, cap, lo, hi, _p(reply), n_slot,
                                                    _p(status), _stream()), "exchange_serve_features")


def exchange_assemble_features(back, F, n_slot, ids, bounds, n_peers, d_n=None, ind_code=None, epoch=0, d_epoch=None,
                               num_ind=0, out=None):
    _chk(back, _f32, "back"); _chk(ids, _i32, "ids"); _chk(bounds, _i32, "bounds"); _chk(ind_code, _i32, "ind_code", True)
    n = ids.numel()
    if back.numel() != n_peers * n_slot * F or bounds.numel() != n_peers + 1:
        raise ValueError("exchange_assemble_features: back / bounds do not match (n_peers, n_slot, F)")
    if out is None:
        out = torch.empty((n, F + num_ind), dtype=_f32, device=ids.device)
    _lib.check(lib().grapes_exchange_assemble_features(_p(back), F, n_slot, _p(ids), n, _p(d_n), _p(bounds), n_peers,
                                                       _p(ind_code), epoch, _p(d_epoch), num_ind, _p(out), _stream()),
               "exchange_assemble_features")
    return out


def exchange_halo_positions(ids, bounds, n_peers, n_slot, d_n=None, ind_code=None, pos=None, code_pos=None):
    """(pos int32[len(ids)], code_pos int32[n_peers * n_slot] | None): where the rows of `ids` sit inside the exchanged buffer
    back[n_peers * n_slot, F], and their indicator words at those positions (see include/grapes_hip.h)."""
    _chk(ids, _i32, "ids"); _chk(bounds, _i32, "bounds"); _chk(ind_code, _i32, "ind_code", True)
    n = ids.numel()
    if pos is None:
        pos = torch.empty(n, dtype=_i32, device=ids.device)
    if ind_code is not None and code_pos is None:
        code_pos = torch.zeros(n_peers * n_slot, dtype=_i32, device=ids.device)
    _lib.check(lib().grapes_exchange_halo_positions(_p(ids), n, _p(d_n), _p(bounds), n_peers, n_slot, _p(ind_code), _p(pos),
                                                    _p(code_pos) if ind_code is not None else None, _stream()), "exchange_halo_positions")
    return pos, (code_pos if ind_code is not None else None)


def exchange_note_rows(loc, ids, pos, base, d_n=None, node_map=None, batch=None, d_n_batch=None, idx_a=None, idx_b=None):
    """loc[ids[i]] = base + pos[row(i)] with row(i) = node_map[ids[i]] (ids that are rows of `batch` only) or idx_b[idx_a[i]]
    (include/grapes_hip.h: grapes_exchange_note_rows) — where a node's feature row sits among the rows already received."""
    _chk(loc, _i32, "loc"); _chk(ids, _i32, "ids"); _chk(pos, _i32, "pos"); _chk(node_map, _i32, "node_map", True)
    _chk(batch, _i32, "batch", True); _chk(idx_a, _i32, "idx_a", True); _chk(idx_b, _i32, "idx_b", True)
    _lib.check(lib().grapes_exchange_note_rows(_p(loc), _p(ids), ids.numel(), _p(d_n), _p(node_map), _p(batch),
                                               batch.numel() if batch is not None else 0, _p(d_n_batch), _p(idx_a), _p(idx_b),
                                               _p(pos), int(base), _stream()), "exchange_note_rows")


# ------------------------------------------------------------------------------- losses + Adam (§8f N2)
def classifier_loss(logits, local_rows, target_ids, labels, out_grad=None):
    """(loss_c [1], d loss_c / d logits [n_rows, C]) — main.py:260,267.  labels: int64 [N] or fp32 [N, C]."""
    _chk(logits, _f32, "logits"); _chk(local_rows, _i32, "local_rows"); _chk(target_ids, _i32, "target_ids")
    n_rows, C = logits.shape
    multi = labels.dim() == 2
    _chk(labels, _f32 if multi else _i64, "labels")
    if out_grad is None:
        out_grad = torch.empty_like(logits)
    loss = torch.empty(1, dtype=_f32, device=logits.device)
    _lib.check(lib().grapes_classifier_loss(_p(logits), n_rows, C, _p(local_rows), _p(target_ids),
                                            None if multi else _p(labels), _p(labels) if multi else None,
                                            local_rows.numel(), _p(out_grad), _p(loss), _stream()), "classifier_loss")
    return loss, out_grad


def gflownet_loss(hop_stats, loss_c, loss_coef, log_z_raw=None, log_z_init=0.0, reinforce=False):
    """out4 = [loss_gfn, grad scale, log_z, sum log-probs] — main.py:272-282."""
    _chk(hop_stats, _f32, "hop_stats"); _chk(loss_c, _f32, "loss_c"); _chk(log_z_raw, _f32, "log_z_raw", True)
    hops, stride = hop_stats.shape
    out = torch.empty(4, dtype=_f32, device=hop_stats.device)
    _lib.check(lib().grapes_gflownet_loss(_p(log_z_raw), float(log_z_init), _p(hop_stats), hops, stride, _p(loss_c),
                                          float(loss_coef), 1 if reinforce else 0, _p(out), _stream()), "gflownet_loss")
    return out


def dropout_fwd(x, p, philox_seed=0, philox_offset=0, d_philox_offset=None, d_n=None):
    """F.dropout(x, p, training=True) on the Philox stream (include/grapes_hip.h): -> (y, keep bytes).  With d_philox_offset the
    device counter is used and advanced."""
    _chk(x, _f32, "x")
    n, f = x.shape
    y = torch.empty_like(x)
    keep = torch.empty((n, f), dtype=torch.uint8, device=x.device)
    _lib.check(lib().grapes_dropout_fwd(_p(x), _p(y), _p(keep), n, _p(d_n), f, float(p), int(philox_seed), int(philox_offset),
                                        _p(d_philox_offset), _stream()), "dropout_fwd")
    return y, keep


def dropout_bwd(dy, keep, p, d_n=None):
    _chk(dy, _f32, "dy")
    n, f = dy.shape
    dx = torch.empty_like(dy)
    _lib.check(lib().grapes_dropout_bwd(_p(dy), _p(keep), _p(dx), n, _p(d_n), f, float(p), _stream()), "dropout_bwd")
    return dx


def logit_var_reg(logits, reg, d_n=None, dlogits=None):
    """reg * sum_r var(logits[r, :]) (main.py:260-261).  Without dlogits: returns the term ([1], for step_losses(loss_extra=));
    with dlogits: adds its gradient to them in place."""
    _chk(logits, _f32, "logits"); _chk(dlogits, _f32, "dlogits", True)
    n, C = logits.shape
    out = torch.empty(1, dtype=_f32, device=logits.device) if dlogits is None else None
    _lib.check(lib().grapes_logit_var_reg(_p(logits), n, _p(d_n), C, float(reg), _p(out), _p(dlogits), _stream()), "logit_var_reg")
    return out if dlogits is None else dlogits


def step_losses(logits, node_map, target_ids, labels, hop_stats, loss_coef, z_out=None, d_nz=None, log_z_init=0.0,
                reinforce=False, many_workgroups=True, loss_extra=None):
    """classifier_loss (target rows = node_map[target_ids]) + mean of z_out + gflownet_loss in one launch
    (main.py:259-282).  Returns (loss_c [1], d loss_c / d logits, out4)."""
    _chk(logits, _f32, "logits"); _chk(node_map, _i32, "node_map"); _chk(target_ids, _i32, "target_ids")
    _chk(hop_stats, _f32, "hop_stats"); _chk(z_out, _f32, "z_out", True)
    n_rows, C = logits.shape
    multi = labels.dim() == 2
    _chk(labels, _f32 if multi else _i64, "labels")
    hops, stride = hop_stats.shape
    dlogits = torch.empty_like(logits)
    loss = torch.empty(1, dtype=_f32, device=logits.device)
    out4 = torch.empty(4, dtype=_f32, device=logits.device)
    ws = _ws(lib().grapes_step_losses_workspace_bytes(target_ids.numel()), logits.device) if many_workgroups else None
    ticket = _ticket(logits.device)[8:9] if many_workgroups else None
    _lib.check(lib().grapes_step_losses(_p(logits), n_rows, C, _p(node_map), _p(target_ids),
                                        None if multi else _p(labels), _p(labels) if multi else None, target_ids.numel(),
                                        _p(dlogits), _p(loss), _p(z_out), 0 if z_out is None else z_out.numel(), _p(d_nz),
                                        float(log_z_init), _p(hop_stats), hops, stride, float(loss_coef),
                                        1 if reinforce else 0, _p(out4), _p(loss_extra), _p(ws), _p(ticket), _stream()), "step_losses")
    return loss, dlogits, out4


class FusedAdam:
    """torch.optim.Adam updates of one or more optimisers in ONE launch (main.py:268,289).  Works on the optimisers'
    own state tensors (exp_avg, exp_avg_sq, step), so state_dict() / checkpoints stay those of torch.optim.Adam.
    Hyper-parameters are read when the descriptor is built; call refresh() after changing them (lr schedules)."""

    def __init__(self, optimizers, mirrors=None):
        """mirrors: {parameter: (w_pad | None, image | None)} — copies of a [rows, K] weight that the launch keeps current with the
        update: w_pad fp32 [rows, >= K] (its padding columns stay as they are) and / or the bf16x3 split image of
        weight_split_image (rows <= 256).  Both must hold the CURRENT weight when the first step runs."""
        self.optimizers = [o for o in optimizers if o is not None]
        self.mirrors = {id(p): (p, m) for p, m in (mirrors or {}).items()}
        self.refresh()

    def refresh(self):
        import struct
        recs, self._keep, dev, maxn = [], [], None, 1
        self._keep_mirrors = []
        for opt in self.optimizers:
            for gp in opt.param_groups:
                if gp.get("amsgrad", False):
                    raise ValueError("FusedAdam: amsgrad is not supported")
                b1, b2 = gp["betas"]
                for p in gp["params"]:
                    if not p.requires_grad:
                        continue
                    _chk(p.data, _f32, "param")
                    dev = p.device
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                    st = opt.state[p]
                    if "step" not in st:                     # same lazy state torch creates (capturable layout)
                        st["step"] = torch.zeros((), dtype=_f32, device=dev)
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    if not (st["step"].is_cuda and st["step"].dtype == _f32):
                        raise ValueError("FusedAdam needs device-resident step counters (Adam(capturable=True))")
                    self._keep.append((p, p.grad, st["exp_avg"], st["exp_avg_sq"], st["step"]))
                    w_pad = img = None
                    K = ld_pad = 0
                    mir = self.mirrors.get(id(p))
                    if mir is not None:
                        w_pad, img = mir[1]
                        if p.dim() != 2 or not p.is_contiguous() or (img is not None and p.shape[0] > 256) or p.numel() >= 2 ** 31:
                            raise ValueError("FusedAdam: a mirrored weight is a contiguous [rows <= 256 for an image, K] matrix")
                        K = int(p.shape[1])
                        if w_pad is not None:
                            _chk(w_pad, _f32, "w_pad")
                            if w_pad.shape[0] != p.shape[0] or w_pad.shape[1] < K or w_pad.stride(1) != 1:
                                raise ValueError("FusedAdam: w_pad is fp32 [rows, >= K]")
                            ld_pad = int(w_pad.stride(0))
                        if img is not None and img.numel() < int(lib().grapes_weight_split_image_bytes(K)):
                            raise ValueError("FusedAdam: the image is smaller than grapes_weight_split_image_bytes(K)")
                        self._keep_mirrors.append((w_pad, img))
                    recs.append(struct.pack("PPPPPqdddddiiPPii", p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(),
                                            st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(), p.numel(), float(gp["lr"]),
                                            float(b1), float(b2), float(gp["eps"]), float(gp.get("weight_decay", 0.0)),
                                            1 if gp.get("maximize", False) else 0, 0,
                                            w_pad.data_ptr() if w_pad is not None else 0, img.data_ptr() if img is not None else 0,
                                            K, ld_pad))
                    maxn = max(maxn, p.numel())
        if not recs:
            raise ValueError("FusedAdam: no parameters")
        if len({k[4].data_ptr() for k in self._keep}) != len(self._keep):
            raise ValueError("FusedAdam: every parameter needs its own step counter (as torch.optim.Adam keeps them)")
        assert len(recs[0]) == lib().grapes_adam_desc_bytes()
        self.n, self.maxn = len(recs), maxn
        self.desc = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(dev)
        self.ticket = torch.zeros(self.n, dtype=_i32, device=dev)

    def step(self, slabs: "DeferredSlabs" = None):
        """slabs: deferred few-row weight-gradient slabs (DeferredSlabs, not flushed) whose sums are formed inside this launch;
        sets whose target is not a gradient of these optimisers (or not of its size) make the whole batch flush first."""
        for p, g, *_ in self._keep:
            if p.grad is not g:
                raise RuntimeError("FusedAdam: a .grad tensor was replaced; gradients must be written in place")
        if slabs is not None and slabs.sets:
            grads = {g.data_ptr(): g.numel() for _, g, *_ in self._keep}
            if len(slabs.sets) <= 8 and all(grads.get(s[2].data_ptr()) == s[3] for s in slabs.sets):
                import ctypes as C
                k = len(slabs.sets)
                sl = (C.c_void_p * k)(*[s[1] for s in slabs.sets])
                gr = (C.c_void_p * k)(*[s[2].data_ptr() for s in slabs.sets])
                _lib.check(lib().grapes_adam_step_slabs(_p(self.desc), self.n, self.maxn, _p(self.ticket), k, sl, gr, slabs.n,
                                                        _p(slabs.d_n), 1 if slabs.acc else 0, _stream()), "adam_step_slabs")
                slabs.sets = []
                return
            slabs.flush()
        _lib.check(lib().grapes_adam_step(_p(self.desc), self.n, self.maxn, _p(self.ticket), _stream()), "adam_step")
