"""ctypes binding of ``libotter_hip.so`` (the C ABI declared in ``include/otter_hip.h``).

There is no fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_long, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OTTER_HIP_LIB") or os.path.join(_HERE, "libotter_hip.so")

ABI_VERSION = 14

DT_BF16, DT_F16, DT_F32 = 0, 1, 2

# epilogue word (mirrors include/otter_hip.h)
ACT_NONE, ACT_QUICKGELU, ACT_GELU, ACT_RELU = 0, 1, 2, 3
OUT_F32 = 4
RESIDUAL = 8
TILE_128 = 16


class OtterHipError(RuntimeError):
    pass


class ReadoutBlock(Structure):
    _fields_ = [(n, c_void_p) for n in (
        "q_ln_w", "q_ln_b", "wq_t", "bq", "wo_t", "bo", "ln_w", "ln_b", "fc1_t", "fc1_b", "fc2_t", "fc2_b")]


class ReadoutProblem(Structure):
    """``otter_readout_problem``: one pooling of a multi-problem readout launch."""
    _fields_ = [("blocks", POINTER(ReadoutBlock)), ("n_blocks", c_int),
                ("query", c_void_p), ("final_ln_w", c_void_p), ("final_ln_b", c_void_p),
                ("kv", c_void_p), ("ldkv", c_long), ("mask", c_void_p), ("out", c_void_p), ("ldo", c_long),
                ("n_obs", c_int), ("nk", c_int), ("readouts", c_int), ("d", c_int), ("heads", c_int), ("hidden", c_int)]


_P, _I, _L, _F = c_void_p, c_int, c_long, c_float

_SIGNATURES = {
    "otter_gemm": [_I, _P, _L, _P, _I, _P, _P, _L, _P, _L, _I, _I, _I, _I, _P],
    "otter_gemm_ws": [_I, _P, _L, _P, _I, _P, _P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _P, _L, _P],
    "otter_gemm_static_ws": [_I, _P, _L, _P, _I, _P, _P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    "otter_gemm_ln": [_I, _P, _L, _P, _I, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    "otter_row_stats": [_P, _I, _L, _P, _I, _I, _F, _P],
    "otter_gemm_residual_stats": [_I, _P, _L, _P, _I, _P, _P, _L, _I, _I, _I, _P, _F, _P, _L, _P],
    "otter_layernorm": [_P, _I, _L, _P, _P, _P, _L, _I, _I, _F, _I, _P],
    "otter_vit_embed_lnpre": [_P, _L, _P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _F, _P],
    "otter_text_embed": [_P, _I, _I, _P, _I, _P, _P, _I, _L, _I, _I, _P],
    "otter_im2col_patches": [_I, _P, _I, _P, _I, _I, _I, _I, POINTER(c_float), POINTER(c_float), _P],
    "otter_attn_fwd_d64": [_I, _P, _L, _P, _L, _I, _I, _I, _I, _P],
    "otter_attn_policy": [_P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "otter_text_aware_pool_fwd": [_I, _P, _L, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "otter_text_aware_pool_fwd_io": [_I, _P, _L, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "otter_pos_emb_pack": [_I, _P, _P, _I, _I, _P],
    "otter_readout_pool_fwd": [POINTER(ReadoutBlock), _I, _P, _P, _P, _P, _L, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "otter_readout_pool_fwd_multi": [POINTER(ReadoutProblem), _I, _P],
    "otter_l2norm_rows": [_I, _P, _L, _P, _L, _I, _I, _P],
    "otter_linear_f32": [_P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    "otter_convert_rows": [_I, _P, _L, _P, _L, _I, _I, _I, _I, _P, _I, _I, _P],
    "otter_repeat_rows": [_P, _L, _P, _L, _I, _I, _I, _P],
    "otter_add_pe": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "otter_add_pe_rows": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "otter_attn_policy_rows": [_P, _L, _P, _L, _P, _L, _P, _P, _I, _I, _I, _I, _I, _P],
    "otter_window_append": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "otter_resize_bicubic_u8": [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _I, _P, _P, _I, _P],
    "otter_action_step": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, c_double, _P],
    # training step
    "otter_sgemm_strided": [_P, _L, _L, _P, _L, _L, _P, _P, _L, _I, _I, _I, _I, _P, _P],
    "otter_act_fwd": [_P, _P, _L, _I, _P],
    "otter_act_bwd": [_P, _P, _P, _L, _I, _P],
    "otter_col_sum": [_P, _L, _P, _L, _I, _I, _P, _I, _P],
    "otter_layernorm_bwd": [_P, _L, _P, _P, _L, _P, _L, _P, _P, _P, _I, _I, _F, _P],
    "otter_small_attn_fwd": [_P, _L, _L, _P, _P, _L, _L, _P, _L, _P, _L, _L, _I, _I, _I, _I, _I, _I, _P],
    "otter_small_attn_bwd": [_P, _L, _L, _P, _P, _L, _L, _P, _L, _P, _L, _L, _P, _L, _L, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "otter_loss_fwd_bwd": [_P, _P, _L, _I, _F, _P, _P, _P],
    "otter_grad_stats": [_P, _L, _P, _P, _P],
    "otter_adamw_step": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P, _P],
    "otter_scale_inplace": [_P, _L, _F, _P],
    "otter_text_aware_pool_temperature_grad": [_I, _P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
}

_lib = None


def load() -> ctypes.CDLL:
    """Load the library once; raise loudly when it is absent or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OtterHipError(
            f"{LIB_PATH} not found: build it with `make -C otter_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU or PyTorch fallback for the device path.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.otter_last_error.restype = c_char_p
    lib.otter_last_error.argtypes = []
    lib.otter_hip_abi_version.restype = c_int
    lib.otter_hip_abi_version.argtypes = []
    got = lib.otter_hip_abi_version()
    if got != ABI_VERSION:
        raise OtterHipError(f"{LIB_PATH}: ABI version {got}, expected {ABI_VERSION}; rebuild the library")
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.otter_grad_stats_workspace_floats.restype = c_long
    lib.otter_grad_stats_workspace_floats.argtypes = []
    lib.otter_pos_emb_packed_elems.restype = c_long
    lib.otter_pos_emb_packed_elems.argtypes = [c_int, c_int]
    lib.otter_gemm_workspace_floats.restype = c_long
    lib.otter_gemm_workspace_floats.argtypes = [c_int]
    lib.otter_gemm_static_ws_max_n.restype = c_long
    lib.otter_gemm_static_ws_max_n.argtypes = []
    lib.otter_gemm_static_ws_init.restype = c_int
    lib.otter_gemm_static_ws_init.argtypes = []
    _lib = lib
    return lib


def call(name: str, *args) -> None:
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise OtterHipError(f"{name} failed (code {rc}): {lib.otter_last_error().decode()}")


def exported_names():
    return sorted(list(_SIGNATURES) + ["otter_last_error", "otter_hip_abi_version", "otter_grad_stats_workspace_floats",
                                            "otter_pos_emb_packed_elems", "otter_gemm_workspace_floats", "otter_gemm_static_ws_max_n",
                                            "otter_gemm_static_ws_init"])
