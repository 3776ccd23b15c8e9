"""ctypes binding of libclusterfusion_hip.so (C-ABI: include/clusterfusion_hip.h).

There is NO fallback: if the shared library is missing or an entry point is absent the import of
the ops fails loudly.  Build it with `python -m clusterfusion_amd.build`.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# CF_LIB_PATH: A/B of two builds of the same library (development only)
LIB_PATH = os.environ.get("CF_LIB_PATH") or os.path.join(_PKG, "libclusterfusion_hip.so")

CF_W_OUT_IN, CF_W_IN_OUT = 0, 1
CF_ROPE_NEOX, CF_ROPE_GPTJ = 0, 1
CF_PROFILE_STAGES = 4
CF_MLA_STAGES = 3
CF_TP_HANDLE_BYTES = 64


class cf_dims(C.Structure):
    _fields_ = [("hidden", C.c_int32), ("n_q_heads", C.c_int32), ("n_kv_heads", C.c_int32),
                ("head_dim", C.c_int32)]


class cf_layer_args(C.Structure):
    _fields_ = [
        ("dims", cf_dims),
        ("batch", C.c_int32), ("weight_layout", C.c_int32), ("rope_style", C.c_int32), ("eps", C.c_float),
        ("x", C.c_void_p), ("residual", C.c_void_p), ("weight_qkv", C.c_void_p), ("weight_o", C.c_void_p),
        ("rms_weight", C.c_void_p),
        ("k_cache", C.c_void_p), ("v_cache", C.c_void_p),
        ("kv_cache_ptrs_k", C.c_void_p), ("kv_cache_ptrs_v", C.c_void_p),
        ("layer_id", C.c_int32), ("page_size", C.c_int32),
        ("seq_len", C.c_int64),
        ("kv_indptr", C.c_void_p), ("kv_indices", C.c_void_p), ("kv_seq_lens", C.c_void_p),
        ("max_seq_len", C.c_int64),
        ("cos", C.c_void_p), ("sin", C.c_void_p), ("positions", C.c_void_p), ("rope_row_stride", C.c_int64),
        ("out", C.c_void_p), ("residual_out", C.c_void_p), ("k_new", C.c_void_p), ("v_new", C.c_void_p),
        ("write_kv_to_cache", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("stream", C.c_void_p),
        ("tp_areas", C.POINTER(C.c_void_p)), ("tp_rank", C.c_int32), ("tp_world", C.c_int32),
    ]


# every symbol include/clusterfusion_hip.h declares: (restype, argtypes)
_P, _I32, _I64, _F, _SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
EXPORTS = {
    "cf_abi_version": (C.c_int, []),
    "cf_last_error": (C.c_char_p, []),
    "cf_workspace_bytes": (_SZ, [C.POINTER(cf_dims), _I32]),
    "cf_algorithmic_bytes": (C.c_uint64, [C.POINTER(cf_dims), _I32, _I64, _I32]),
    "cf_decoder_layer_ex": (C.c_int, [C.POINTER(cf_layer_args)]),
    "cf_llama_decoder_layer": (C.c_int, [_P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "cf_llama_decoder_layer_out_in": (C.c_int, [_P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "cf_llama_decoder_layer_sglang": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _P, _F, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "cf_llama_decoder_layer_batch_decode_sglang": (
        C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _F, _P, _P, _I32, _I64, _P, _SZ, _P]),
    "cf_rmsnorm": (C.c_int, [_P, _P, _P, _F, _I32, _I32, _P, _P, _P]),
    "cf_deepseek_workspace_bytes": (_SZ, []),
    "cf_deepseek_algorithmic_bytes": (C.c_uint64, [_I64, _I32]),
    "cf_deepseek_decoder_layer": (C.c_int, [_P] * 9 + [_I64] + [_P] * 4 + [_F, _I32, _P, _P, _P, _SZ, _P]),
    "cf_deepseek_profile_enable": (C.c_int, [_I32]),
    "cf_deepseek_profile_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(_I64), _I32]),
    "cf_profile_enable": (C.c_int, [_I32]),
    "cf_profile_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(_I64), _I32]),
    "cf_set_tuning": (C.c_int, [_I32]),
    "cf_set_path": (C.c_int, [_I32]),
    "cf_last_path": (C.c_int, []),
    "cf_last_variant": (C.c_char_p, []),
    "cf_take_sticky_error": (C.c_uint32, []),
    "cf_debug_occupy": (C.c_int, [_P, _I32, _I32, _I64]),
    "cf_relayout_weights": (C.c_int, [C.POINTER(cf_dims), _P, _P, _P, _P, _P]),
    "cf_debug_set_trace": (C.c_int, [_P]),
    "cf_debug_set_flags": (C.c_int, [_I32]),
    "cf_workspace_init": (C.c_int, [_P, _SZ, _P]),
    "cf_workspace_status": (C.c_int, [_P, _P, C.POINTER(C.c_uint32)]),
    "cf_workspace_last_arm": (C.c_int, [_P, _P, C.POINTER(C.c_uint32)]),
    "cf_tp_oneshot_bytes": (_SZ, [_I32, _I32]),
    "cf_tp_area_alloc": (C.c_int, [_SZ, C.POINTER(C.c_void_p)]),
    "cf_tp_area_free": (C.c_int, [_P]),
    "cf_tp_area_export": (C.c_int, [_P, _P]),
    "cf_tp_area_import": (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    "cf_tp_area_unmap": (C.c_int, [_P]),
    "cf_tp_area_status": (C.c_int, [_P, _P, C.POINTER(C.c_uint32)]),
    "cf_tp_oneshot_allreduce": (C.c_int, [_P, _P, _I32, _I32, _I32, C.POINTER(C.c_void_p), _I32, _P]),
    "cf_tp_gather": (C.c_int, [_P, _I32, _I32, _I32, C.POINTER(C.c_void_p), _P]),
    "cf_rmsnorm_tp_gather": (C.c_int, [C.POINTER(C.c_void_p), _I32, _I32, _P, _P, _F, _I32, _P, _P, _P, _P]),
    "cf_tp_area_clear_error": (C.c_int, [_P, _P]),
}

_lib = None


def load():
    """Load the library once and type every entry point.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"clusterfusion_amd: {LIB_PATH} not found -- the HIP extension is not built "
            "(run `python -m clusterfusion_amd.build`); there is no CPU/eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        try:
            fn = getattr(lib, name)          # AttributeError if the symbol is absent
        except AttributeError:
            if os.environ.get("CF_LIB_PATH"):   # A/B against an older build (development only): newer entry points are absent
                continue
            raise
        fn.restype, fn.argtypes = res, args
    if lib.cf_abi_version() != 2:
        raise RuntimeError("clusterfusion_amd: ABI version mismatch")
    _lib = lib
    return lib


class CFError(RuntimeError):
    pass


def check(rc: int):
    if rc != 0:
        msg = load().cf_last_error().decode()
        raise CFError(f"libclusterfusion_hip error {rc}: {msg}")
