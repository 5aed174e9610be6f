"""ctypes binding of libumbrella_hip.so (the C ABI in include/umbrella_hip.h).

The product path has NO CPU fallback: if the HIP library is missing, or a call
returns non-zero, this raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"``.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UMB_LIB_PATH") or os.path.join(_HERE, "csrc", "libumbrella_hip.so")   # override: experiments only

F16, BF16 = 0, 1


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float16:
        return F16
    raise ValueError(f"umbrella_amd computes in fp16 or bf16, got {dt}")


class UmbLinear(C.Structure):
    _fields_ = [("w", C.c_void_p), ("meta", C.c_void_p), ("N", C.c_int32), ("K", C.c_int32),
                ("awq", C.c_int32), ("R", C.c_int32), ("S", C.c_int32), ("tb", C.c_int32), ("S_row", C.c_int32),
                ("pad_", C.c_int32), ("w_rows", C.c_void_p)]


class UmbLayer(C.Structure):
    _fields_ = [("qkv", UmbLinear), ("o", UmbLinear), ("gu", UmbLinear), ("down", UmbLinear),
                ("norm1", C.c_void_p), ("norm2", C.c_void_p), ("qkv_bias", C.c_void_p)]


class UmbModel(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("L", C.c_int32), ("H", C.c_int32), ("I", C.c_int32),
                ("Hq", C.c_int32), ("Hkv", C.c_int32), ("D", C.c_int32), ("V", C.c_int32),
                ("Lmax", C.c_int32), ("pad_", C.c_int32), ("eps", C.c_float), ("attn_scale", C.c_float),
                ("embed", C.c_void_p), ("lm_head", UmbLinear), ("final_norm", C.c_void_p),
                ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p), ("k_cache", C.c_void_p),
                ("vt_cache", C.c_void_p), ("layers", C.POINTER(UmbLayer))]


class UmbWorkspace(C.Structure):
    _fields_ = [("h", C.c_void_p), ("xn", C.c_void_p), ("q", C.c_void_p), ("attn", C.c_void_p),
                ("act", C.c_void_p), ("partial", C.c_void_p), ("attn_po", C.c_void_p), ("attn_ml", C.c_void_p),
                ("pos", C.c_void_p), ("slot", C.c_void_p), ("prefix", C.c_void_p), ("logits", C.c_void_p),
                ("hw", C.c_void_p), ("ssq", C.c_void_p), ("counters", C.c_void_p), ("attn_counters", C.c_void_p),
                ("Tmax", C.c_int32), ("attn_chunk", C.c_int32), ("attn_splits", C.c_int32), ("ssq_stride", C.c_int32),
                ("fused", C.c_int32), ("defer_norm", C.c_int32), ("chain_xchg", C.c_void_p)]


class UmbChain(C.Structure):
    _fields_ = [("w_o", C.c_void_p), ("w_gu", C.c_void_p), ("w_down", C.c_void_p), ("w_qkv", C.c_void_p),
                ("attn", C.c_void_p), ("h", C.c_void_p), ("hw", C.c_void_p), ("ssq", C.c_void_p),
                ("norm2", C.c_void_p), ("next_norm", C.c_void_p), ("pos", C.c_void_p), ("slot", C.c_void_p),
                ("cosT", C.c_void_p), ("sinT", C.c_void_p), ("q_out", C.c_void_p), ("k_cache", C.c_void_p),
                ("vt_cache", C.c_void_p), ("xchg", C.c_void_p),
                ("T", C.c_int32), ("Tmax", C.c_int32), ("front", C.c_int32), ("tail", C.c_int32), ("H", C.c_int32),
                ("I", C.c_int32), ("NQKV", C.c_int32), ("ssq_stride", C.c_int32), ("ssq_groups_in", C.c_int32),
                ("Hq", C.c_int32), ("Hkv", C.c_int32), ("D", C.c_int32), ("Lmax", C.c_int32), ("eps", C.c_float)]


class UmbGemmFused(C.Structure):
    _fields_ = [("ssq_in", C.c_void_p), ("ssq_groups", C.c_int32), ("ssq_dim", C.c_float), ("eps", C.c_float),
                ("pad0", C.c_int32), ("counters", C.c_void_p), ("h", C.c_void_p), ("hw", C.c_void_p),
                ("norm_w", C.c_void_p), ("ssq_out", C.c_void_p), ("ssq_out_stride", C.c_int32), ("pad1", C.c_int32),
                ("pos", C.c_void_p), ("slot", C.c_void_p), ("cosT", C.c_void_p), ("sinT", C.c_void_p),
                ("q_out", C.c_void_p), ("k_cache", C.c_void_p), ("vt_cache", C.c_void_p), ("Hq", C.c_int32),
                ("Hkv", C.c_int32), ("D", C.c_int32), ("Lmax", C.c_int32)]


class UmbGemmLL(C.Structure):
    _fields_ = [("row_from", C.c_int32), ("round_out", C.c_int32), ("ssq_in", C.c_void_p), ("ssq_groups", C.c_int32),
                ("ssq_in_stride", C.c_int32), ("ssq_dim", C.c_float), ("eps", C.c_float), ("h", C.c_void_p),
                ("hw", C.c_void_p), ("norm_w", C.c_void_p), ("ssq_out", C.c_void_p), ("ssq_out_stride", C.c_int32),
                ("pad0", C.c_int32), ("pos", C.c_void_p), ("slot", C.c_void_p), ("cosT", C.c_void_p),
                ("sinT", C.c_void_p), ("q_out", C.c_void_p), ("k_cache", C.c_void_p), ("vt_cache", C.c_void_p),
                ("bias", C.c_void_p), ("Hq", C.c_int32), ("Hkv", C.c_int32), ("D", C.c_int32), ("Lmax", C.c_int32)]


class UmbStep(C.Structure):
    _fields_ = [("T", C.c_int32), ("tree_off", C.c_int32), ("tokens", C.c_void_p), ("positions", C.c_void_p),
                ("slots", C.c_void_p), ("prefix_len", C.c_void_p), ("tokens_all", C.c_void_p),
                ("n_ptr", C.c_void_p), ("depth", C.c_void_p), ("mask_bits", C.c_void_p),
                ("mask_words", C.c_int32), ("n_mask_keys", C.c_int32), ("head_from", C.c_int32),
                ("layer_begin", C.c_int32), ("layer_end", C.c_int32), ("skip_embed", C.c_int32)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


TP_MAX_RANKS = 16


class UmbTPPeer(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("slot", C.c_void_p * TP_MAX_RANKS),
                ("flag", C.c_void_p * TP_MAX_RANKS), ("epoch", C.c_void_p), ("arrive", C.c_void_p), ("status", C.c_void_p),
                ("cap", C.c_int64), ("spin_limit", C.c_int64)]


class UmbTP(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("allreduce", ALLREDUCE_FN), ("ctx", C.c_void_p),
                ("peer", C.POINTER(UmbTPPeer)), ("peer_max_floats", C.c_int64)]


MAX_SLABS = 8


class UmbOffload(C.Structure):
    _fields_ = [("host_slabs", C.POINTER(C.c_void_p)), ("slab_bytes", C.c_size_t), ("dev_slab", C.c_void_p * MAX_SLABS),
                ("copy_stream", C.c_void_p), ("ev_copied", C.c_void_p * MAX_SLABS), ("ev_free", C.c_void_p * MAX_SLABS),
                ("prefetched", C.POINTER(C.c_int32)), ("n_slabs", C.c_int32), ("pad_", C.c_int32)]


_P, _I, _F = C.c_void_p, C.c_int, C.c_float
# name -> argtypes (all return int unless listed in _VOID / _STR)
SIGNATURES = {
    "umb_repack_dense": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "umb_awq_repack": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "umb_gemm_plan": [_I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I)],
    "umb_gemm_plan2": [_I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)],
    "umb_gemm_wide_split": [_I, _I, _I],
    "umb_gemm": [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "umb_gemm_fused": [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, C.POINTER(UmbGemmFused), _I, _P],
    "umb_gemm_ll": [_P, _P, _P, _P, _I, _I, _I, _I, _I, C.POINTER(UmbGemmLL), _I, _P],
    "umb_gemv": [_P, _P, _P, _I, _I, _I, _I, C.POINTER(UmbGemmLL), _I, _P],
    "umb_gemv_ok": [_I, _I, _I, _I],
    "umb_gemv_groups": [_I, _I, _I],
    "umb_repack_rows": [_P, _P, _I, _I, _I, _I, _I, _P],
    "umb_chain_ok": [_I, _I, _I, _I, _I, _I],
    "umb_chain_xchg_bytes": [_I, _I, _I],
    "umb_chain_xchg_init": [_P, _I, _I, _I, _P],
    "umb_chain_status": [_P, _I, _I, _I, C.POINTER(C.c_uint32), _P],
    "umb_draft_chain": [C.POINTER(UmbChain), _I, _P],
    "umb_head_stream_ok": [_I, _I, _I],
    "umb_head_stream": [_P, _P, _P, _I, _I, _F, _P, _I, _I, _I, _I, _I, _P],
    "umb_ll_plan": [_I, _I, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)],
    "umb_ll_token_tiles": [_I],
    "umb_to_fm": [_P, _P, _I, _I, _I, _P],
    "umb_from_fm": [_P, _P, _I, _I, _I, _P],
    "umb_embed_ll": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "umb_tree_attn2": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _I, _I, _P],
    "umb_rmsnorm": [_P, _P, _P, _F, _I, _I, _I, _P],
    "umb_rmsnorm_fm": [_P, _P, _P, _F, _I, _I, _I, _I, _P],
    "umb_reduce_residual_norm": [_P, _I, _I, _I, _P, _P, _P, _P, _F, _I, _P],
    "umb_reduce_residual_norm_fm": [_P, _I, _I, _I, _P, _P, _P, _P, _F, _I, _I, _P],
    "umb_reduce_residual_hw": [_P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "umb_reduce_silu_mul": [_P, _I, _I, _I, _P, _I, _P],
    "umb_reduce_qkv_rope": [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P],
    "umb_stream_read": [_P, C.c_size_t, _P, _P],
    "umb_rope_inplace": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "umb_kv_append": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "umb_h2d_layer": [_P, _P, C.c_size_t, _P, _P, _P],
    "umb_reduce_qkv_rope2": [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _F, _F, _I, _P],
    "umb_embed_prep": [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "umb_embed_prep_fm": [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "umb_tree_attn": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _I, _P],
    "umb_argmax_rows": [_P, _P, _I, _I, _P],
    "umb_topk_rows": [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P],
    "umb_topk_rows_ws": [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, C.c_size_t, _P],
    "umb_sample_rows": [_P, _P, _I, _I, _P, _P, _F, _F, _I, _F, _P, _I, _P, _P, _P],
    "umb_sample_rows_uniform": [_P, _P, _I, _I, _P, _P, _F, _F, _I, _F, _P, _I, _I, _I, _P, _P, _P],
    "umb_beam_expand": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P],
    "umb_accept_scan": [_P, _P, _P, _P, _I, _P, _I, _P, _P, _P],
    "umb_kv_compact": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "umb_set_int": [_P, _I, _P],
    "umb_mask_eos": [_P, _P, _I, _P],
    "umb_write_token": [_P, _P, _P, _P],
    "umb_apply_override": [_P, _P, _P, _P, _I, _I, _P],
    "umb_model_forward": [C.POINTER(UmbModel), C.POINTER(UmbWorkspace), C.POINTER(UmbStep), _P],
    "umb_model_forward_tp": [C.POINTER(UmbModel), C.POINTER(UmbWorkspace), C.POINTER(UmbStep), C.POINTER(UmbTP), _P],
    "umb_sum_splits": [_P, _I, C.c_int64, _P],
    "umb_tp_publish": [C.POINTER(UmbTPPeer), _P, _I, C.c_int64, _P],
    "umb_tp_reduce_residual_norm": [C.POINTER(UmbTPPeer), _I, _I, _P, _P, _P, _P, _F, _I, _P],
    "umb_tp_xchg_alloc": [C.c_size_t, C.POINTER(C.c_void_p), _P, C.POINTER(C.c_int)],
    "umb_tp_xchg_open": [_P, C.POINTER(C.c_void_p)],
    "umb_tp_xchg_close": [_P],
    "umb_tp_xchg_free": [_P],
    "umb_model_forward_offload": [C.POINTER(UmbModel), C.POINTER(UmbWorkspace), C.POINTER(UmbStep),
                                  C.POINTER(UmbOffload), _P],
    "umb_bench_launch": [_I, _I, _P, _P],
    "umb_version": [],
}
_VOID = {"umb_gemm_plan", "umb_gemm_plan2", "umb_ll_plan"}
_STR = {"umb_version"}
_SIZE = {"umb_chain_xchg_bytes"}

_lib = None


def load() -> C.CDLL:
    """dlopen the in-tree HIP library (never a fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run __graft_entry__.build() "
            "(hipcc --offload-arch=gfx950). umbrella_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if the ABI is incomplete
        fn.argtypes = args
        fn.restype = None if name in _VOID else (C.c_char_p if name in _STR else (C.c_size_t if name in _SIZE else C.c_int))
    _lib = lib
    return lib


class UmbError(RuntimeError):
    pass


def check(rc: int, what: str = ""):
    if rc != 0:
        raise UmbError(f"libumbrella_hip {what} failed with code {rc}")


def stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


def call(name: str, *args):
    lib = load()
    conv = []
    for a in args:
        if isinstance(a, torch.Tensor) or a is None:
            conv.append(ptr(a))
        elif isinstance(a, C.Structure):
            conv.append(C.byref(a))
        else:
            conv.append(a)
    rc = getattr(lib, name)(*conv, stream_ptr())
    check(rc, name)
