"""ctypes binding of libgrapes_hip.so (the C-ABI declared in include/grapes_hip.h).

This is the stub a maintainer of the (pure-Python) reference would add to call the MI355X path;
see INTEGRATION.md.  There is NO CPU fallback: if the shared library is missing or a symbol is
absent, importing/using the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DIAG_LIB_PATH = os.path.join(_HERE, "libgrapes_hip_diag.so")
# The product library has ONE configuration and reads no environment switch.  GRAPES_DIAG=1 (profiles/, the tests that compare
# forms) selects the diagnostic build — the same kernels plus the A/B switches and probes — and turns the Python-side
# switches on (diag_switch below).  GRAPES_LIB_PATH overrides the path (other diagnostic builds: stamps, lb768).
DIAG = os.environ.get("GRAPES_DIAG", "0") == "1"
LIB_PATH = os.environ.get("GRAPES_LIB_PATH") or (DIAG_LIB_PATH if DIAG else os.path.join(_HERE, "libgrapes_hip.so"))
ABI_MAJOR, ABI_MINOR = 3, 2          # include/grapes_hip.h: GRAPES_ABI_VERSION = 100 * MAJOR + MINOR


def diag_switch(name: str, default: str) -> str:
    """Value of an A/B switch: the environment variable in a diagnostic session (GRAPES_DIAG=1), else ALWAYS the default —
    the product path has one configuration."""
    if os.environ.get("GRAPES_DIAG", "0") == "1":
        return os.environ.get(name, default)
    return default


P = C.c_void_p
I32 = C.c_int32
I64 = C.c_int64
U32 = C.c_uint32
U64 = C.c_uint64
F32 = C.c_float
SZ = C.c_size_t

# name -> (restype, [argtypes])   — order and meaning exactly as in include/grapes_hip.h
SIGNATURES = {
    "grapes_abi_version": (I32, []),
    "grapes_target_arch": (C.c_char_p, []),
    "grapes_build_flavor": (C.c_char_p, []),
    "grapes_linear_gathered_workspace_bytes": (C.c_size_t, [I32, I32, I32]),
    "grapes_linear_fwd_gathered": (I32, [P, I32, I32, P, P, U32, P, I32, P, P, I32, P, I32, P, P]),
    "grapes_linear_bwd_weight_gathered": (I32, [P, P, I32, I32, P, P, U32, P, I32, U32, P, I32, P, I32, I32, P, P]),
    "grapes_split_gathered_available": (I32, [I32]),
    "grapes_weight_split_image_bytes": (C.c_size_t, [I32]),
    "grapes_weight_split_image": (I32, [P, I32, I32, I32, P, P]),
    "grapes_weight_split_image_padded": (I32, [P, I32, I32, I32, P, P, I32, P]),
    "grapes_weight_split_images": (I32, [I32, P, P, P, P, P, P, P, P]),
    "grapes_linear_bwd_weight_gathered_split_ld": (I32, [P, P, I32, I32, P, P, U32, P, I32, U32, P, I32, I32, P, I32, I32, P, P]),
    "grapes_linear_fwd_gathered_split": (I32, [P, I32, I32, P, P, U32, P, I32, P, P, I32, P, I32, P]),
    "grapes_linear_fwd_gathered_split_tail_workspace_bytes": (C.c_size_t, [I32]),
    "grapes_linear_fwd_gathered_split_tail": (I32, [P, I32, I32, P, I32, P, U32, P, P, P, P, I32, P, I32, P, P]),
    "grapes_linear_fwd_gathered_split_k": (I32, [P, I32, I32, P, P, U32, P, I32, P, P, I32, P, I32, P, P]),
    "grapes_linear_fwd_gathered_split_k_workspace_bytes": (C.c_size_t, [I32, I32, I32]),
    "grapes_linear_bwd_weight_gathered_split_workspace_bytes": (C.c_size_t, [I32, I32]),
    "grapes_linear_bwd_weight_gathered_split_multi_available": (I32, [I32, I32]),
    "grapes_linear_bwd_weight_gathered_split_multi_workspace_bytes": (C.c_size_t, [I32, I32, I32]),
    "grapes_linear_bwd_weight_gathered_split_multi": (I32, [I32, P, P, I32, I32, P, P, U32, P, P, P, P, P, P, P, I32, P, P, P]),
    "grapes_linear_bwd_weight_gathered_split": (I32, [P, P, I32, I32, P, P, U32, P, I32, U32, P, I32, P, I32, I32, P, P]),
    "grapes_csr_build_workspace_bytes": (C.c_size_t, [I64, I32]),
    "grapes_csr_build": (I32, [P, P, I64, I32, P, P, P, P, P, P]),
    "grapes_kernel_clock_enable": (I32, [P, I64]),
    "grapes_kernel_clock_launches": (I32, []),
    "grapes_kernel_clock_entry": (I32, [I32, P, P, P]),
    "grapes_kernel_clock_rate_khz": (I32, []),
    "grapes_gcn_aggregate_fwd_rec": (I32, [P, P, P, P, P, P, P, I32, P, I32, I32, P, P, P, P]),
    "grapes_scatter_rows": (I32, [P, I64, P, P, I64, I32, I32, P, I32, I32, P]),
    "grapes_rider_record_begin": (I32, []),
    "grapes_rider_record_end": (I32, []),
    "grapes_rider_count": (I32, [I32]),
    "grapes_rider_attach": (I32, [I32, I32, P]),
    "grapes_rider_release": (I32, [P]),
    "grapes_rider_detach": (I32, [P, P]),
    "grapes_rider_launch": (I32, [I32, P]),
    "grapes_rider_free": (I32, [I32]),
    "grapes_graph_chain_create": (I32, [P, I32, I32, P, P]),
    "grapes_graph_chain_launch": (I32, [P, P]),
    "grapes_graph_chain_destroy": (I32, [P]),
    "grapes_tensormap_update": (I32, [P, P, I32, P, P]),
    "grapes_tensormap_map": (I32, [P, P, P, I64, P, P]),
    "grapes_frontier_offsets": (I32, [P, P, I32, P, P, P, P]),
    "grapes_frontier_expand": (I32, [P, P, P, I32, P, P, I32, P, P, P, P, P]),
    "grapes_frontier_expand_fused": (I32, [P, P, P, I32, P, I32, P, P, P, P, P, P, P, I32, P, P, P, P, P]),
    "grapes_frontier_expand_fused_ext": (I32, [P, P, P, I32, P, I32, P, P, P, P, P, P, P, I32, P, P, P, P, P, P, P, P, P]),
    "grapes_slice_stage_words": (SZ, [I32]),
    "grapes_bitmap_mark": (I32, [P, P, P, I64, P, I32, P, P]),
    "grapes_bitmap_mark_rows": (I32, [P, P, P, I32, P, P, I32, P, P]),
    "grapes_bitmap_clear": (I32, [P, P, I64, P, P]),
    "grapes_frontier_compact_workspace_bytes": (SZ, [I32, I32]),
    "grapes_frontier_compact": (I32, [P, P, P, I32, I32, P, P, P, P, P, P, U32, P, I32, P, P, SZ, P, SZ, P, SZ, P, P, P, P, P]),
    "grapes_frontier_compact_counted": (I32, [P, P, P, I32, I32, P, P, P, P, P, P, U32, P, I32, P, P, SZ, P, SZ, P, SZ, P, P, P, P, P, P]),
    "grapes_gcn_prepare_counted": (I32, [P, P, P, I32, P, P, I32, P, P, P, P, P, P, P, P, P, P, P, P, P, I64, I32, P, P]),
    "grapes_bitmap_mark_hop": (I32, [P, P, P, P, I32, P, P, P, I32, P, I32, P, P]),
    "grapes_bitmap_mark_lists": (I32, [P, P, P, I32, P, P, I32, P, P, I32, P, P, I32, P, I32, P, P, P]),
    "grapes_union_sorted": (I32, [P, I32, P, P, I32, P, P, I32, P, P, I32, P, I32, I32, P, P, P, P, P, P]),
    "grapes_slice_mark": (I32, [P, P, I32, P, I32, P, P]),
    "grapes_slice_remark": (I32, [P, P, I32, P, P, I32, P, P, P, I32, P, P]),
    "grapes_slice_filter_workspace_bytes": (SZ, [I32]),
    "grapes_slice_filter": (I32, [P, P, P, I32, P, I32, P, P, P, P, P, I32, P, P]),
    "grapes_indicator_mark": (I32, [P, P, I32, P, U32, P, I32, I32, P]),
    "grapes_step_begin": (I32, [P, P, I32, P, I32, P, I32, I32, I32, P, P, I32, I32, P, P]),
    "grapes_gather_rows": (I32, [P, I32, P, I32, P, P, U32, P, I32, P, P]),
    "grapes_gcn_prepare_workspace_bytes": (SZ, [I32, I32]),
    "grapes_gcn_prepare_zero_words": (SZ, [I32]),
    "grapes_gcn_long_items_capacity": (I32, [I32]),
    "grapes_gcn_prepare": (I32, [P, P, I32, P, P, I32, P, I32, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "grapes_gcn_prepare_prefetching": (I32, [P, P, I32, P, P, I32, P, I32, P, P, P, P, P, P, P, P, P, P, P, P, P, I64, I32, P]),
    "grapes_gcn_prepare_small_batch": (I32, [I32, P, P, P, P, P, I32, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "grapes_gcn_prepare_from_csr": (I32, [P, I32, P, P, P, I32, P]),
    "grapes_linear_fwd": (I32, [P, P, P, I32, P, I32, I32, P]),
    "grapes_linear_fwd_row_scaled": (I32, [P, P, P, P, I32, P, I32, I32, P]),
    "grapes_eval_predict": (I32, [P, I32, I32, P, P, I32, P, P, P, P]),
    "grapes_linear_bwd_weight_workspace_bytes": (SZ, [I32, I32, I32]),
    "grapes_linear_bwd_weight": (I32, [P, P, P, I32, P, I32, I32, I32, P, P]),
    "grapes_linear_bwd_input": (I32, [P, P, P, I32, P, I32, I32, P]),
    "grapes_linear_bias_act_fwd": (I32, [P, P, P, I32, P, I32, P, I32, I32, P]),
    "grapes_linear_bias_act_head_fwd": (I32, [P, P, P, I32, P, P, P, I32, P, I32, I32, P]),
    "grapes_split_gemm_available": (I32, [I32, I32, I32]),
    "grapes_sampler_head_bwd_multi_workspace_bytes": (SZ, []),
    "grapes_sampler_head_bwd_multi_phase": (I32, [I32, P, P, P, P, P, P, P, P, P, P, P, P, I32, P, P, P, I32, P]),
    "grapes_linear_bias_act_head_fwd_strided": (I32, [P, I32, P, P, I32, P, P, P, I32, P, I32, I32, P]),
    "grapes_linear_bwd_weight_gated_strided": (I32, [P, P, I32, P, I32, P, P, P, P, P, I32, I32, I32, P, P]),
    "grapes_linear_bwd_weight_gated_workspace_bytes": (SZ, [I32, I32, I32]),
    "grapes_linear_bwd_weight_gated": (I32, [P, P, P, P, P, I32, P, I32, I32, I32, P, P, P, P, P]),
    "grapes_linear_bwd_weight_gated_multi": (I32, [I32, P, P, P, P, P, P, P, P, P, I32, I32, I32, P, P]),
    "grapes_linear_bwd_weight_slabs_bytes": (SZ, [I32, I32, I32]),
    "grapes_linear_bwd_weight_slabs": (I32, [P, P, P, I32, P, I32, I32, I32, P, P]),
    "grapes_linear_bwd_weight_slabs_and_input": (I32, [P, P, P, P, P, I32, P, I32, I32, I32, P, P]),
    "grapes_slab_reduce_sets": (I32, [I32, P, P, P, I32, P, I32, P]),
    "grapes_gcn_aggregate_narrow_pair": (I32, [P, P, P, P, P, P, P, P, P, I32, P, P]),
    "grapes_linear_relu_head_fwd_bits": (I32, [P, I32, P, P, P, P, P, I32, P, I32, I32, P]),
    "grapes_linear_relu_head_fwd_bits_pair": (I32, [P, I32, P, P, P, P, P, P, I32, P, P, P, P, P, I32, I32, P, I32, I32, P]),
    "grapes_linear_bwd_weight_bits_multi_cols": (I32, [I32, P, P, P, P, P, P, P, P, P, P, I32, P, P, I32, I32, I32, P, P]),
    "grapes_linear_bwd_weight_bits_pair_cols": (I32, [I32, P, P, P, P, P, P, P, P, P, P, I32, P, P, I32, P, P, P, P, P, P, I32, I32, I32, P, P]),
    "grapes_gcn_aggregate_gather_fwd": (I32, [P, I32, I32, P, P, U32, P, I32, P, P, P, P, P, I32, P, P]),
    "grapes_gcn_aggregate_gather_fwd_peers": (I32, [P, P, I32, I32, I32, P, P, U32, P, I32, P, P, P, P, P, I32, P, P]),
    "grapes_peer_export": (I32, [P, P, P]),
    "grapes_peer_open": (I32, [P, U64, P]),
    "grapes_peer_close": (I32, [P, U64]),
    "grapes_peer_copy": (I32, [P, P, SZ, P]),
    "grapes_gcn_aggregate_workspace_bytes": (SZ, [I32, I32]),
    "grapes_gcn_aggregate_fwd": (I32, [P, P, P, P, P, P, I32, P, I32, I32, P, P, I32, P, P]),
    "grapes_gcn_aggregate_fwd_head": (I32, [P, P, P, P, P, P, I32, P, I32, I32, P, P, P, P]),
    "grapes_gcn_aggregate_fwd_prescaled": (I32, [P, P, P, P, P, P, I32, P, I32, I32, P, P, I32, P, P]),
    "grapes_scale_rows": (I32, [P, P, P, I64, I32, P]),
    "grapes_gcn_aggregate_bwd_rank1_workspace_bytes": (SZ, [I32, I32]),
    "grapes_gcn_aggregate_bwd_rank1": (I32, [P, P, P, P, P, P, P, P, P, I32, I32, P, I32, P, P, I32, P, P]),
    "grapes_gcn_aggregate_bwd_rank1_bits": (I32, [P, P, P, P, P, P, P, P, P, P, I32, I32, P, I32, P, P, I32, P, P]),
    "grapes_gcn_aggregate_bwd_rank1_bits_multi": (I32, [I32, P, P, P, P, P, P, P, P, P, P, P, P, P, I32, P, P, P, P, P]),
    "grapes_gcn_aggregate_bwd_workspace_bytes": (SZ, [I32, I32]),
    "grapes_gcn_aggregate_bwd": (I32, [P, P, P, P, P, P, P, P, I32, I32, P, I32, P, P, I32, P, P, P]),
    "grapes_sampler_workspace_bytes": (SZ, [I32]),
    "grapes_gumbel_topk": (I32, [P, P, P, U64, U64, P, I32, P, I32, I32, P, P, P, P, P, P, P, P, P, I32, P, P, P, P]),
    "grapes_gumbel_topk_hist": (I32, [P, P, P, U64, U64, P, I32, P, I32, I32, P, P, P, P, P, P, P, P, P, I32, P, P, P, P, P]),
    "grapes_gumbel_topk_deferred_ext": (I32, [P, P, P, U64, U64, P, I32, P, I32, I32, P, P, P, P, P, P, P, P, P, I32, P, P, P, P, P, P, P, P, P]),
    "grapes_sampler_hist_words": (I32, []),
    "grapes_gumbel_topk_from_aggregate": (I32, [P, P, P, P, P, P, I32, P, P, P, P, U64, U64, P, I32, P, I32, I32, P, P, P, P, P, P, P, P, P, I32, P, P, P, P]),
    "grapes_bernoulli_logprob_bwd": (I32, [P, P, P, P, P, P, I32, P, P, I32, P, P, P]),
    "grapes_philox_uniform": (I32, [P, I64, U64, U64, P]),
    "grapes_reduce_sum": (I32, [P, I32, P, I32, P, P]),
    "grapes_fill": (I32, [P, I32, P, F32, P, F32, P, I32, P]),
    "grapes_classifier_loss": (I32, [P, I32, I32, P, P, P, P, I32, P, P, P]),
    "grapes_gflownet_loss": (I32, [P, F32, P, I32, I32, P, F32, I32, P, P]),
    "grapes_step_losses_workspace_bytes": (SZ, [I32]),
    "grapes_step_losses": (I32, [P, I32, I32, P, P, P, P, I32, P, P, P, I32, P, F32, P, I32, I32, F32, I32, P, P, P, P, P]),
    "grapes_logit_var_reg": (I32, [P, I32, P, I32, F32, P, P, P]),
    "grapes_dropout_fwd": (I32, [P, P, P, I32, P, I32, F32, U64, U64, P, P]),
    "grapes_dropout_bwd": (I32, [P, P, P, I32, P, I32, F32, P]),
    "grapes_adam_desc_bytes": (I32, []),
    "grapes_adam_step": (I32, [P, I32, I64, P, P]),
    "grapes_adam_step_slabs": (I32, [P, I32, I64, P, I32, P, P, I32, P, I32, P]),
    "grapes_exchange_pack_query": (I32, [P, I32, P, I32, P, P]),
    "grapes_exchange_serve_rows": (I32, [P, P, P, I32, I32, I32, I32, P, I64, I32, P, P, P]),
    "grapes_exchange_recv_rows": (I32, [P, I64, P, I32, P, P, I32, I32, P, P, P, P, P, P, P]),
    "grapes_exchange_serve_features": (I32, [P, I32, P, I32, I32, I32, I32, P, I32, P, P]),
    "grapes_exchange_assemble_features": (I32, [P, I32, I32, P, I32, P, P, I32, P, U32, P, I32, P, P]),
    "grapes_exchange_halo_positions": (I32, [P, I32, P, P, I32, I32, P, P, P, P]),
    "grapes_exchange_note_rows": (I32, [P, P, I32, P, P, P, I32, P, P, P, P, I32, P]),
}

_lib = None


# measurement-only entry points: exported by the diagnostic build only (include/grapes_hip.h, #ifdef GRAPES_DIAG)
DIAG_SIGNATURES = {
    "grapes_debug_gemm_fwd": (I32, [P, P, P, I32, I32, I32, I32, P]),
    "grapes_debug_gather_probe": (I32, [P, I32, P, P, I32, I32, I32, I32, P]),
    "grapes_debug_tsplit_fwd": (I32, [P, I32, I32, P, P, P, I32, I32, I32, P]),
    "grapes_debug_tsplit_dw": (I32, [P, P, I32, I32, P, I32, I32, P, I32, P]),
    "grapes_feature_planes_bytes": (C.c_size_t, [I64, I32]),
    "grapes_feature_split_planes": (I32, [P, I64, I32, P, P]),
    "grapes_feature_planes_register": (I32, [P, I32, P]),
}


class GrapesHipError(RuntimeError):
    pass


_diag_lib = None


def _bind(lib, table, what):
    for name, (res, args) in table.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise GrapesHipError(f"{what} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args


def load_diag():
    """The diagnostic build (make -C grapes_amd/csrc diag): every product entry point plus DIAG_SIGNATURES.  Tests and
    profiles/ that need a measurement-only entry point load THIS; the product path never does."""
    global _diag_lib
    if _diag_lib is not None:
        return _diag_lib
    if _lib is not None and LIB_PATH == DIAG_LIB_PATH:
        _diag_lib = _lib
        return _diag_lib
    if not os.path.exists(DIAG_LIB_PATH):
        raise GrapesHipError(f"{DIAG_LIB_PATH} is missing: build it with `make -C grapes_amd/csrc diag`")
    lib = C.CDLL(DIAG_LIB_PATH)
    _bind(lib, SIGNATURES, "libgrapes_hip_diag.so")
    _bind(lib, DIAG_SIGNATURES, "libgrapes_hip_diag.so")
    if lib.grapes_build_flavor() != b"diag":
        raise GrapesHipError("libgrapes_hip_diag.so was not built with -DGRAPES_DIAG")
    _diag_lib = lib
    return lib


def load():
    """Loads the shared library (once).  Raises GrapesHipError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GrapesHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C grapes_amd/csrc`).  grapes_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    _bind(lib, SIGNATURES, os.path.basename(LIB_PATH))
    if lib.grapes_build_flavor() == b"diag":
        _bind(lib, DIAG_SIGNATURES, os.path.basename(LIB_PATH))
    v = lib.grapes_abi_version()
    if v // 100 != ABI_MAJOR or v % 100 < ABI_MINOR:
        raise GrapesHipError(f"{os.path.basename(LIB_PATH)}: ABI {v // 100}.{v % 100}, this binding needs {ABI_MAJOR}.>={ABI_MINOR}")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "misaligned pointer"}.get(rc, f"hipError {rc}")
        raise GrapesHipError(f"{what} failed: {kind}")
