"""ctypes binding of libgnm.so (include/gnm.h).  The product path has NO fallback: if the
library is missing or a symbol is absent this raises, it never routes around the HIP code."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GNM_LIBRARY") or os.path.join(_HERE, "libgnm.so")   # GNM_LIBRARY: A/B against another build (tools)

_lib = None
ABI_VERSION = 7     # GNM_ABI_VERSION of include/gnm.h

_p = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int
_f32 = C.c_float
_sz = C.c_size_t
_pi = C.POINTER(C.c_int)

# name -> (restype, argtypes); must list every symbol include/gnm.h declares
SIGNATURES = {
    "gnm_abi_version": (_i32, []),
    "gnm_last_error": (C.c_char_p, []),
    "gnm_num_cus": (_i32, []),
    "gnm_max_partial_blocks": (_i32, []),
    "gnm_graph_build_index": (_i32, [_p, _p, _i64, _i64, _p, _p, _p, _p, _p, _p, _p]),
    "gnm_graph_edge_locality": (_i32, [_p, _p, _i64, _i64, _i64, C.POINTER(C.c_double)]),
    "gnm_graph_locality_order": (_i32, [_p, _p, _i64, _i64, _p, _p, C.POINTER(C.c_double)]),
    "gnm_gemm_f32_workspace_bytes": (_sz, [_i32, _i64, _i64, _i64]),
    "gnm_gemm_f32": (_i32, [_i32, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _i32, _p, _sz, _p]),
    "gnm_gemm_tn_colsum_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "gnm_gemm_tn_colsum": (_i32, [_i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _sz, _p]),
    "gnm_colsum_workspace_bytes": (_sz, [_i64, _i64]),
    "gnm_colsum_f32": (_i32, [_i64, _i64, _p, _i64, _p, _p, _sz, _p]),
    "gnm_gather_rows_f32": (_i32, [_i64, _i64, _p, _p, _p, _p]),
    "gnm_relu_mask_f32": (_i32, [_i64, _p, _p, _p]),
    "gnm_bn_finalize": (_i32, [_p, _i32, _i64, _i32, _p, _p, _f32, _p, _p]),
    "gnm_bn_bwd_finalize": (_i32, [_p, _i32, _i64, _i32, _p, _p, _p, _p]),
    "gnm_edge_t_stats_fwd": (_i32, [_i64, _i32, _p, _p, _p, _p, _p, _pi, _p]),
    "gnm_edge_gate_fwd": (_i32, [_i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "gnm_node_agg_src_fwd": (_i32, [_i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _pi, _p]),
    "gnm_node_update_fwd": (_i32, [_i64, _i32, _p, _p, _p, _p, _p]),
    "gnm_node_bwd_stats": (_i32, [_i64, _i32, _p, _p, _p, _p, _pi, _p]),
    "gnm_node_bwd_apply": (_i32, [_i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "gnm_edge_bwd_dst": (_i32, [_i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _pi, _p]),
    "gnm_edge_bwd_src": (_i32, [_i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p]),
    "gnm_edge_bwd_gt": (_i32, [_i64, _i32, _p, _p, _p, _p, _p, _p, _p]),
    "gnm_ln_edge_gate_fwd": (_i32, [_i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p]),
    "gnm_ln_node_update_fwd": (_i32, [_i64, _i32, _p, _p, _p, _p, _p, _i32, _p]),
    "gnm_ln_node_bwd": (_i32, [_i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _pi, _i32, _p]),
    "gnm_ln_edge_bwd_dst": (_i32, [_i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _pi, _i32, _p]),
    "gnm_ln_edge_bwd_src": (_i32, [_i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p]),
    "gnm_rowtile_workspace_bytes": (_sz, [_i32]),
    "gnm_set_occupancy_cap": (_i32, [_i32]),
    "gnm_debug_set_variant": (_i32, [C.c_char_p, _i32]),
    "gnm_set_matmul_mode": (_i32, [_i32]),
    "gnm_get_matmul_mode": (_i32, []),
    "gnm_edge_t_fused_fwd": (_i32, [_i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _pi, _p, _sz, _p]),
    "gnm_edge_bwd_gt_nn": (_i32, [_i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gnm_node_proj_fwd": (_i32, [_i64, _i32, _i32, _p, _p, _p, _p, _p, _sz, _p]),
    "gnm_node_proj_bwd_workspace_bytes": (_sz, [_i32]),
    "gnm_node_proj_bwd": (_i32, [_i64, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gnm_node_proj_bwd_nn": (_i32, [_i64, _i32, _i32, _p, _p, _p, _p, _p, _sz, _p]),
    "gnm_node_proj_bwd_tn": (_i32, [_i64, _i32, _i32, _p, _p, _p, _p, _p, _p, _sz, _i32, _p]),
    "gnm_node_proj_bwd_nn_stats": (_i32, [_i64, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _pi, _p, _sz, _p]),
    "gnm_tn128_bgrad": (_i32, [_i64, _i32, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gnm_edge_bwd_chain": (_i32, [_i64, _i64, _i32] + [_p] * 24 + [_pi, _p, _sz, _p]),
    "gnm_edge_bwd_chain_src": (_i32, [_i64, _i64, _i32] + [_p] * 24 + [_p, _i64, _p] + [_pi, _p, _sz, _p]),
    "gnm_edge_bwd_top": (_i32, [_i64, _i64, _i32] + [_p] * 15 + [_p, _i64, _p, _pi, _p, _sz, _p]),
    "gnm_edge_bwd_src_fix": (_i32, [_i64, _p, _i64, _i64, _i32] + [_p] * 10 + [_p]),
    "gnm_node_bgrad": (_i32, [_i64, _i32] + [_p] * 8 + [_i64, _p, _p]),
    "gnm_graph_build_sweep_plan": (_i32, [_p, _p, _p, _i64, _i64, _i64, _i32, _i32, _i64, _p, _p, _p, C.POINTER(C.c_int64), _pi]),
    "gnm_graph_build_sweep_plan_device": (_i32, [_p, _p, _p, _i64, _i64, _i64, _i32, _i32, _i64, _p, _p, _p, _p, _p, _p, _p, _p]),
    "gnm_sweep_partition": (_i32, [_i64, _i32, C.POINTER(C.c_int64), _pi]),
    "gnm_edge_gate2_fwd": (_i32, [_i64, _i64, _i32] + [_p] * 9 + [_i64, _i64] + [_p] * 11 + [_pi, _p]),
    "gnm_ln_edge_gate2_fwd": (_i32, [_i64, _i64, _i32] + [_p] * 4 + [_i32] + [_p] * 6 + [_i64, _i64] + [_p] * 11 + [_pi, _p]),
    "gnm_edge_bwd_fused_workspace_bytes": (_sz, []),
    "gnm_edge_bwd_fused": (_i32, [_i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gnm_edge_bwd_fused_gt": (_i32, [_i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gnm_ln_edge_bwd_top": (_i32, [_i64, _i64, _i32, _p, _p, _p, _p, _p, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _sz, _p]),
    "gnm_ln_edge_bwd_chain": (_i32, [_i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64,
                                     _p, _p, _sz, _p]),
    "gnm_ln_edge_bwd_src_fix": (_i32, [_i64, _p, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p]),
    "gnm_edge_encoder_fwd": (_i32, [_i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p]),
    "gnm_edge_encoder_bwd_workspace_bytes": (_sz, []),
    "gnm_edge_encoder_bwd": (_i32, [_i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gnm_predictor_score_fwd": (_i32, [_i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "gnm_predictor_score_bwd": (_i32, [_i64, _i32, _p, _p, _p, _p, _p, _pi, _p]),
    "gnm_predictor_fused_workspace_bytes": (_sz, []),
    "gnm_predictor_fused_fwd": (_i32, [_i64, _i32, _i32, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gnm_predictor_fused_bwd": (_i32, [_i64, _i32, _i32, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _sz, _p]),
    "gnm_tn128_workspace_bytes": (_sz, []),
    "gnm_tn128": (_i32, [_i64, _p, _i64, _i32, _p, _p, _p, _p, _p, _sz, _p]),
    "gnm_compose_workspace_bytes": (_sz, [_i32]),
    "gnm_compose_partials_doubles": (_sz, []),
    "gnm_layer_forward": (_i32, [_p, _i32, _p, _p, _p, _p]),                          # struct pointers: see include/gnm.h
    "gnm_stack_backward": (_i32, [_p, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "gnm_decode_build_adjacency": (_i32, [_p, _p, _i64, _i64, _p, _p, _p, _p, _p, _p]),
    "gnm_decode_iteration": (_i64, [_i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _i32, _p, _i64, _p]),
    "gnm_reduce_partials": (_i32, [_p, _i32, _i32, _i32, _p, _p]),
    "gnm_seg_sum_rows": (_i32, [_i64, _i32, _p, _p, _p, _p, _i64, _p]),
    "gnm_pagerank_pe_workspace_bytes": (_sz, [_i64]),
    "gnm_pagerank_pe": (_i32, [_i64, _i64, _p, _p, _p, _i32, C.c_double, _p, _p, _sz, _p]),
    "gnm_edge_feats_zscore": (_i32, [_i64, _p, _p, _p, _p, _sz, _p]),
    "gnm_bce_fwd_bwd": (_i32, [_i64, _p, _p, _f32, _p, _p, _p, _sz, _p]),
}


class GnmError(RuntimeError):
    pass


def load():
    """dlopen libgnm.so and bind every symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GnmError(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise GnmError(f"libgnm.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    v = lib.gnm_abi_version()
    if v != ABI_VERSION:
        raise GnmError(f"libgnm.so ABI version {v}, expected {ABI_VERSION} (rebuild: python __graft_entry__.py)")
    _lib = lib
    mode = os.environ.get("GNM_MATMUL", "").strip().lower()
    if mode:                                     # matmul mode of the fused kernels (gnm.h); library default: f16x2
        if mode not in MATMUL_MODES:
            raise GnmError(f"GNM_MATMUL={mode!r}: expected one of {sorted(MATMUL_MODES)}")
        check(lib.gnm_set_matmul_mode(MATMUL_MODES[mode]), "gnm_set_matmul_mode")
    for kv in filter(None, os.environ.get("GNM_VARIANTS", "").split(",")):   # kernel-generation A/B (gnm_debug_set_variant)
        k, _, v = kv.partition("=")
        check(lib.gnm_debug_set_variant(k.strip().encode(), int(v)), "gnm_debug_set_variant")
    return lib


MATMUL_MODES = {"f32": 0, "fp32": 0, "bf16x3": 1, "f16x2": 2}
DEFAULT_MATMUL_MODE = "f16x2"


def set_matmul_mode(mode: str) -> None:
    """'f16x2' (default: two fp16 terms of a power-of-two multiple, 22 significand bits per operand, three MFMAs per product, in
    every split-mode matrix kernel), 'bf16x3' (exact 3-way bf16 split, six bf16 MFMAs per product, fp32
    accumulate) or 'f32' (fp32 MFMA)."""
    if mode not in MATMUL_MODES:
        raise GnmError(f"matmul mode {mode!r}: expected one of {sorted(MATMUL_MODES)}")
    check(load().gnm_set_matmul_mode(MATMUL_MODES[mode]), "gnm_set_matmul_mode")


def get_matmul_mode() -> str:
    return ("f32", "bf16x3", "f16x2")[load().gnm_get_matmul_mode()]


def split_mode() -> bool:
    """True in the matmul modes that split fp32 operands into 16-bit terms (bf16x3, f16x2): the kernels built only for those."""
    return load().gnm_get_matmul_mode() >= 1


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().gnm_last_error().decode(errors="replace")
        raise GnmError(f"{what or 'libgnm'} failed (rc={rc}): {msg}")


def csrc_sha() -> str:
    """sha256 (16 hex digits) over the library's sources (csrc/* and include/gnm.h, sorted by name): what
    profiles/*_traffic.json records beside the PMC bytes, so that bench.py can tell a stale pass from a current one."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".cpp")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "gnm.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
