"""ctypes binding of ``libsat_hip.so`` (C ABI declared in ``include/sat_hip.h``).

The product has NO CPU fallback: importing this module without the built library, or
calling an op with non-HIP tensors, raises.  Build with
``python __graft_entry__.py build`` (or ``make -C friendly-stable-audio-tools_amd/csrc``).
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libsat_hip.so")


class SatError(RuntimeError):
    pass


ABI_VERSION = 6          # include/sat_hip.h: sat_version()


class SatDitCfg(Structure):
    _fields_ = [("io_channels", c_int32), ("embed_dim", c_int32), ("depth", c_int32), ("num_heads", c_int32),
                ("cond_token_dim", c_int32), ("cond_embed_dim", c_int32), ("global_cond_dim", c_int32),
                ("max_seq_len", c_int32), ("adaln", c_int32), ("gemm_dtype", c_int32), ("fp8_families", c_int32), ("ln_fold", c_int32),
                ("cross_attention", c_int32), ("tile_policy", c_int32)]


class SatT5Cfg(Structure):
    _fields_ = [("vocab_size", c_int32), ("d_model", c_int32), ("d_kv", c_int32), ("d_ff", c_int32), ("num_layers", c_int32),
                ("num_heads", c_int32), ("rel_buckets", c_int32), ("rel_max_distance", c_int32), ("gated_gelu", c_int32),
                ("proj_dim", c_int32), ("eps", c_float)]


class SatOobleckCfg(Structure):
    _fields_ = [("is_decoder", c_int32), ("io_channels", c_int32), ("channels", c_int32), ("latent_dim", c_int32),
                ("n_blocks", c_int32), ("c_mults", c_int32 * 8), ("strides", c_int32 * 8), ("gemm_dtype", c_int32)]


_SIGNATURES = {
    "sat_version": (c_int32, []),
    "sat_last_error": (c_char_p, []),
    "sat_dit_plan_create": (c_int32, [POINTER(SatDitCfg), POINTER(c_void_p)]),
    "sat_dit_plan_create_sized": (c_int32, [POINTER(SatDitCfg), c_size_t, POINTER(c_void_p)]),
    "sat_dit_plan_destroy": (None, [c_void_p]),
    "sat_dit_plan_set_tensor": (c_int32, [c_void_p, c_char_p, c_void_p, c_int64]),
    "sat_dit_plan_finalize": (c_int32, [c_void_p, c_void_p]),
    "sat_dit_workspace_bytes": (c_int32, [c_void_p, c_int32, c_int32, POINTER(c_size_t)]),
    "sat_dit_prepare_context": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "sat_dit_set_null_context_from": (c_int32, [c_void_p, c_int32]),
    "sat_dit_forward": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_size_t, c_void_p]),
    "sat_dit_denoise_cfg": (c_int32, [c_void_p, c_void_p, c_float, c_float, c_float, c_void_p, c_int32, c_int32, c_void_p,
                                      c_size_t, c_void_p]),
    "sat_dit_profile": (c_int32, [c_void_p, c_int32]),
    "sat_dit_profile_read": (c_int32, [c_void_p, POINTER(ctypes.c_double), POINTER(c_int32), POINTER(c_int64), POINTER(c_int64),
                                       POINTER(c_int64)]),
    "sat_dit_debug": (c_int32, [c_void_p, c_int32]),
    "sat_dit_debug_read": (c_int32, [c_void_p, POINTER(c_float), c_int32, c_void_p]),
    "sat_cfg_combine": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_float, c_void_p]),
    "sat_quant_rows_fp8": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "sat_layernorm_fp8": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "sat_gemm_fp8_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                   c_int32, c_void_p]),
    "sat_quant_mx_rows_fp8": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "sat_gemm_mxfp8_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                     c_int32, c_void_p]),
    "sat_lincomb": (c_int32, [c_void_p, c_void_p, c_float, c_void_p, c_float, c_void_p, c_float, c_void_p, c_float, c_void_p, c_float,
                              c_int64, c_void_p]),
    "sat_dpm_error_partials": (c_int32, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_int64, c_void_p, c_int32, c_void_p]),
    "sat_inpaint_mix": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int64, c_int32, c_void_p]),
    "sat_dpmpp3m_update": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float,
                                     c_float, c_int64, c_void_p]),
    "sat_oobleck_plan_create": (c_int32, [POINTER(SatOobleckCfg), POINTER(c_void_p)]),
    "sat_oobleck_plan_destroy": (None, [c_void_p]),
    "sat_oobleck_plan_set_tensor": (c_int32, [c_void_p, c_char_p, c_void_p, c_int64]),
    "sat_oobleck_plan_finalize": (c_int32, [c_void_p, c_void_p]),
    "sat_oobleck_workspace_bytes": (c_int32, [c_void_p, c_int32, c_int32, POINTER(c_size_t)]),
    "sat_oobleck_decode": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_size_t, c_void_p]),
    "sat_oobleck_encode": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_size_t, c_void_p]),
    "sat_vae_sample": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "sat_float_to_int16": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "sat_layernorm_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "sat_cast_bf16": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    "sat_cross_attention_fused_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                                 c_int32, c_int32, c_void_p]),
    "sat_resample_sinc": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "sat_gemm_bf16_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                    c_void_p]),
    "sat_gemm_f32_workspace_bytes": (c_int32, [c_int32, c_int32, c_int32, c_int32, POINTER(c_size_t)]),
    "sat_gemm_bf16_f32_ws": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                       c_void_p, c_size_t, c_void_p]),
    "sat_gemm_resid_ln_bf16_ws": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                            c_void_p, c_size_t, c_void_p]),
    "sat_gemm_swiglu_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                       c_int32, c_void_p]),
    "sat_attention_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                     c_int32, c_int32, c_void_p]),
    "sat_attention_prescaled_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                               c_int32, c_int32, c_void_p]),
    "sat_qkv_rope_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                    c_int32, c_int32, c_int32, c_void_p]),
    "sat_gemm_resid_ln_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                         c_void_p]),
    "sat_gemm_swiglu_ln_bf16": (c_int32, [c_void_p] * 9 + [c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "sat_qkv_rope_ln_bf16": (c_int32, [c_void_p] * 12 + [c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "sat_t5_plan_create": (c_int32, [c_void_p, c_void_p]),
    "sat_t5_plan_destroy": (None, [c_void_p]),
    "sat_t5_plan_set_tensor": (c_int32, [c_void_p, c_char_p, c_void_p, c_int64]),
    "sat_t5_plan_finalize": (c_int32, [c_void_p, c_void_p]),
    "sat_t5_workspace_bytes": (c_int32, [c_void_p, c_int32, c_int32, c_void_p]),
    "sat_t5_encode": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_size_t, c_void_p]),
    "sat_t5_relative_buckets": (c_int32, [c_int32, c_int32, c_int32, c_void_p]),
    "sat_snake_beta": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "sat_overlap_add": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "sat_number_embed": (c_int32, [c_void_p, c_int32, c_float, c_float, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
}

# the same unit-level entry points on IEEE fp16 operands (gemm_dtype = 3): identical signatures
for _n in ("sat_layernorm_bf16", "sat_cast_bf16", "sat_gemm_bf16_f32", "sat_gemm_swiglu_bf16", "sat_attention_bf16", "sat_cross_attention_fused_bf16", "sat_attention_prescaled_bf16", "sat_qkv_rope_bf16", "sat_gemm_resid_ln_bf16", "sat_gemm_swiglu_ln_bf16", "sat_qkv_rope_ln_bf16", "sat_gemm_bf16_f32_ws", "sat_gemm_resid_ln_bf16_ws"):
    _SIGNATURES[_n.replace("bf16", "f16")] = _SIGNATURES[_n]

_lib = None


def exported_symbols():
    return sorted(_SIGNATURES)


def lib():
    """Loads libsat_hip.so (once).  Raises SatError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SatError(f"HIP extension not built: {LIB_PATH} is missing (run `python __graft_entry__.py build`); "
                           "this package has no CPU fallback")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        if handle.sat_version() != ABI_VERSION:          # the cfg structs below mirror exactly one version of include/sat_hip.h
            raise SatError(f"{LIB_PATH} reports ABI version {handle.sat_version()}, this package binds version {ABI_VERSION}: rebuild "
                           "(python __graft_entry__.py build)")
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().sat_last_error()
        raise SatError(f"libsat_hip error {rc}: {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a contiguous HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise SatError("libsat_hip ops need tensors on a HIP device (torch device 'cuda'); there is no CPU path")
    if not t.is_contiguous():
        raise SatError("libsat_hip ops need contiguous tensors")
    return c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)
