"""ctypes binding of libeav_hip.so (the C ABI declared in include/eav_hip.h).

There is deliberately NO fallback: if the shared library is missing or a symbol
cannot be resolved the import of the calling module fails loudly, and every
call checks the integer status and raises ``EavError`` with the library's own
message.  PyTorch is used only to own device memory and streams; tensors cross
the boundary as raw device pointers (``Tensor.data_ptr()``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (EAV_LIB_PATH: kernel-tuning runs load an alternative build of the same library - tools/ only, never set by the package)
LIB_PATH = os.environ.get("EAV_LIB_PATH") or os.path.join(_HERE, "libeav_hip.so")


class EavError(RuntimeError):
    pass


_p = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_u64 = C.c_uint64
_f = C.c_float
_d = C.c_double

# name -> argtypes (all return an int status)
SIGNATURES = {
    "eav_reduce_partials": [_p, _i, _i64, _i, _f, _p, _p],
    "eav_bn_finalize": [_p, _i, _i, _d, _p, _p, _p, _p, _i, _f, _f, _p, _p, _p, _p, _p],
    "eav_bn_bwd_finalize": [_p, _i, _i, _d, _i, _p, _p, _p, _p, _p],
    "eav_reduce_and_bn_bwd_finalize": [_p, _i, _i64, _i, _p, _p, _i, _i, _d, _i, _p, _p, _p, _p, _p, _i, _i, _d, _i, _p, _p, _p,
                                       _p, _p],
    "eav_renorm_rows": [_p, _i, _i, _f, _p],
    "eav_renorm_rows2": [_p, _i, _i, _p, _i, _i, _f, _p],
    "eav_eegnet_step_prologue": [_p, _p, _p, _p, _p, _p, _p, _p],
    "eav_eegnet_fir_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "eav_eegnet_fir_wgrad": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "eav_eegnet_fir_fwd_indexed": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "eav_eegnet_fir_fwd_fft": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "eav_eegnet_fir_wgrad_fft": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "eav_eegnet_fir_wgrad_indexed": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "eav_eegnet_dw_fwd": [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    "eav_eegnet_dw_fwd_pool_eval": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "eav_eegnet_dw_bwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "eav_eegnet_dw_bwd_fused": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _u64, _p, _p, _p],
    "eav_eegnet_dw_bwd_fused_eval": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _u64, _p, _p, _p],
    "eav_bn_elu_pool_fwd": [_p, _p, _p, _i, _i, _i, _i, _f, _u64, _p, _p, _p],
    "eav_bn_elu_pool_bwd_reduce": [_p, _p, _p, _p, _i, _i, _i, _i, _f, _u64, _p, _p, _p],
    "eav_bn_elu_pool_bwd_apply": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _u64, _p, _p, _p],
    "eav_bn_elu_pool_bwd_eval": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _u64, _p, _p, _p],
    "eav_conv64_prep_weights": [_p, _p, _p, _p],
    "eav_conv64_fwd": [_p, _p, _p, _p, _i, _i, _i, _p],
    "eav_conv64_wgrad": [_p, _p, _p, _i, _i, _i, _p],
    "eav_conv64_fft_fwd": [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    "eav_conv64_fft_wgrad": [_p, _p, _p, _i, _i, _p],
    "eav_dense_softmax_fwd": [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    "eav_dense_softmax_bwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "eav_ce_fwd_bwd": [_p, _p, _p, _p, _p, _p, _i, _i, _p],
    "eav_scale_by_scalar": [_p, _p, _i64, _p],
    "eav_adam_step": [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _i64, _i, _p, _p],
    "eav_counter_inc": [_p, _p],
    "eav_counter_inc4": [_p, _p, _p, _p, _p],
    "eav_step_begin": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p],
    "eav_gather_rows": [_p, _p, _p, _i, _i64, _p],
    "eav_gather_i64": [_p, _p, _p, _i, _p],
    "eav_gemm_f32": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64, _f, _p,
                     _i, _p, _p, _i, _i, _p],
    "eav_gemm_f32_splitk": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "eav_sp_absmax": [_p, _i, _i, _i64, _p, _p],
    "eav_sp_convert": [_p, _i, _i, _i64, _p, _p, _p, _p],
    "eav_sp_convert_gelu": [_p, _i, _i, _i64, _p, _p, _p, _p],
    "eav_sp_convert_colsum": [_p, _i, _i, _i64, _p, _p, _p, _p, _p],
    "eav_layernorm_fwd_amax": [_p, _p, _p, _p, _p, _p, _i, _i, _f, _p, _p],
    "eav_layernorm_bwd_amax": [_p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _p, _p],
    "eav_layernorm_bwd_planes": [_p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p],
    "eav_layernorm_bwd_bound": [_p, _p, _p, _p, _p, _i, _i, _p, _p],
    "eav_gelu_bwd_amax": [_p, _p, _i64, _p, _p],
    "eav_gemm_sp": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i64, _i64, _f, _p, _i, _p, _p, _i, _i, _p, _p],
    "eav_gemm_sp_x1": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i64, _i64, _f, _p, _i, _p, _p, _i, _i, _p, _p],
    "eav_gemm_sp_splitk_x1": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "eav_gemm_sp_splitk_x2": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "eav_gemm_sp_ex": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i64, _i64, _f, _p, _i, _p, _p, _i, _i, _p, _p, _p, _p, _i, _p],
    "eav_colnorm_max": [_p, _i, _i, _i64, _p, _p],
    "eav_norm_max_multi": [_p, _i, _i, _p],
    "eav_sp_bound_scale": [_p, _p, _p, _f, _p],
    "eav_gemm_sp_planes": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i64, _i64, _f, _p, _i, _p, _p, _i, _i, _p, _p, _p, _p],
    "eav_layernorm_fwd_planes": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p],
    "eav_sp_refresh_planes": [_p, _i, _i, _i, _p],
    "eav_rownorm_max": [_p, _i, _i, _i64, _p, _p],
    "eav_tf_forward_scales": [_p, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _i64, _i, _i, _i, _p],
    "eav_gemm_sp_splitk": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "eav_attn_sp_prep": [_p, _p, _p, _p, _i, _i, _i, _i, C.c_uint, _p],
    "eav_attn_fwd_sp": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "eav_tf_forward_scales_qkv": [_p, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _i64, _i, _i, _i, _i, _p],
    "eav_attn_fwd_sp_planes": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "eav_attn_bwd_sp": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "eav_attn_bwd_sp_planes": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "eav_attn_dqkv_bound": [_p, _p, _p, _i, _f, _p],
    "eav_attn_fwd": [_p, _p, _p, _i, _i, _i, _i, _f, _p],
    "eav_attn_bwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "eav_layernorm_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _f, _p],
    "eav_layernorm_bwd": [_p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _p],
    "eav_softmax_fwd": [_p, _i64, _i, _i, _p],
    "eav_softmax_bwd": [_p, _p, _i64, _i, _i, _p],
    "eav_gelu_bwd": [_p, _p, _i64, _p],
    "eav_colsum": [_p, _p, _i, _i, _i, _p],
    "eav_im2col": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "eav_embed_finish": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "eav_embed_bwd": [_p, _p, _p, _i, _i, _i, _i, _p],
    "eav_token_rows": [_p, _p, _i, _i, _i, _i, _i, _p],
    "eav_pair_mean": [_p, _p, _i, _i, _i, _p],
    "eav_ast_fbank": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _d, _d, _f, _f, _p],
    "eav_decimate_fir_f64": [_p, _p, _p, _i, _i64, _i64, _i, _i, _i, _p],
    "eav_sosfilt_f64": [_p, _p, _p, _p, _p, _p, _p, _i, _i64, _i, _i, _p],
    "eav_eegnet_block1_infer": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "eav_tconv_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "eav_tconv_wgrad": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "eav_spatial_fwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "eav_spatial_bwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "eav_dconv_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "eav_dconv_wgrad": [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    "eav_sepconv_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "eav_pointwise_bwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "eav_dwt_bwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "eav_shallow_embed_fwd": [_p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p],
    "eav_shallow_embed_bwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "eav_relu_dropout": [_p, _i64, _f, _u64, _p, _p, _p],
    "eav_relu_dropout_bwd": [_p, _p, _i64, _f, _p],
    "eav_dropout_add": [_p, _p, _p, _i64, _f, _u64, _p, _p, _p],
    "eav_add_strided": [_p, _i, _p, _i, _p, _i, _i64, _i, _p],
    "eav_colstats": [_p, _p, _i64, _i, _i, _p],
    "eav_sqpool_log_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _f, _u64, _p, _p, _p],
    "eav_sqpool_log_bwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _f, _u64, _p, _p, _p],
    "eav_bn_rows_bwd": [_p, _p, _p, _p, _i64, _i, _p],
    "eav_peak_mfma_f32": [_p, _i, _i, _p],
    "eav_peak_copy": [_p, _p, _i64, _p],
    "eav_peak_mfma_f16": [_p, _i, _i, _p],
    "eav_peak_copy_variant": [_p, _p, _i64, _i, _i, _p],
    "eav_peak_l2_read": [_p, _i, _i, _i, _i, _p, _p],
    "eav_resize_normalize_u8": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _d, _p, _p, _p],
}
# helpers that return a plain value (no status)
PLAIN = {
    "eav_abi_version": ([], _i),
    "eav_last_error": ([], C.c_char_p),
    "eav_eegnet_fir_fwd_nparts": ([_i, _i, _i], _i),
    "eav_eegnet_fir_fwd_fft_nparts": ([_i, _i, _i], _i),
    "eav_eegnet_fir_wgrad_fft_ws_floats": ([_i, _i, _i], _i64),
    "eav_eegnet_fir_fft_max_taps": ([], _i),
    "eav_eegnet_fir_wgrad_nparts": ([_i, _i, _i], _i),
    "eav_conv64_ntiles": ([_i], _i),
    "eav_tconv_fwd_nparts": ([_i, _i, _i, _i, _i], _i),
    "eav_shallow_embed_nparts": ([_i, _i], _i),
    "eav_colstats_nparts": ([_i64], _i),
    "eav_tconv_wgrad_nparts": ([_i, _i, _i, _i, _i], _i),
    "eav_spatial_nparts": ([_i, _i], _i),
    "eav_dconv_fwd_nparts": ([_i, _i], _i),
    "eav_sepconv_fwd_nparts": ([_i, _i], _i),
    "eav_pointwise_bwd_nparts": ([_i, _i], _i),
    "eav_conv64_fwd_nparts": ([_i, _i], _i),
    "eav_conv64_fft_nparts": ([_i, _i], _i),
    "eav_conv64_fft_ws_floats": ([_i, _i], _i64),
    "eav_conv64_wgrad_nparts": ([_i, _i], _i),
    "eav_layernorm_bwd_nparts": ([_i], _i),
    "eav_gemm_f32_splitk_plan": ([_i, _i, _i], _i),
    "eav_colsum_nparts": ([_i], _i),
    "eav_sp_kpad": ([_i], _i),
    "eav_sp_convert_colsum_nparts": ([_i], _i),
    "eav_attn_sp_npad": ([_i], _i),
    "eav_gemm_sp_splitk_plan": ([_i, _i, _i], _i),
}

EXPORTS = sorted(list(SIGNATURES) + list(PLAIN))

# test / tuning overrides (include/eav_hip_tuning.h): process-global, never called by the package
TUNING = {"eav_gemm_sp_set_tile": [_i], "eav_gemm_sp_set_splitk": [_i], "eav_sp_set_convert_blocks": [_i],
          "eav_attn_sp_set_nw4_above": [_i]}

# Comparison-only kernels (include/eav_hip_extras.h, `make BENCH_EXTRAS=1` -> libeav_extras.so): bench.py's literal-bf16 leg.
EXTRAS_PATH = os.path.join(_HERE, "libeav_extras.so")
EXTRAS = {
    "eav_gemm_bf16": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64, _f, _p,
                      _i, _p, _p, _i, _i, _p],
    "eav_gemm_bf16_splitk": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
}
_extras = None

_lib = None


def load():
    """Load (once) and type the shared library.  Raises EavError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EavError(
            f"{LIB_PATH} is missing - the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C eav_amd/csrc`). "
            "There is no CPU fallback.")
    # torch ships its own libamdhip64.so.7; import it first so that this library binds to the SAME
    # HIP runtime instance (streams and device pointers are shared with torch's allocator).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _i
    for name, (args, res) in PLAIN.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    for name, args in TUNING.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _i
    _lib = lib
    return lib


def ptr(t):
    """Raw pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def have_extras():
    return os.path.exists(EXTRAS_PATH)


def load_extras():
    global _extras
    if _extras is None:
        if not have_extras():
            raise EavError(f"{EXTRAS_PATH} is missing - {sorted(EXTRAS)} are comparison-only kernels: build them with "
                           "`make -C eav_amd/csrc BENCH_EXTRAS=1`")
        load()
        lib = C.CDLL(EXTRAS_PATH)
        for name, args in EXTRAS.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = _i
        lib.eav_last_error.restype = C.c_char_p
        _extras = lib
    return _extras


# bench.py / tools: set to a dict to have EVERY library call bracketed by HIP events on the current stream
# ({entry point: [(start, end), ...]}); None (default) costs one comparison per call.  Never set by the package itself.
TRACE = None


def call(name, *args):
    lib = load_extras() if name in EXTRAS else load()
    if TRACE is not None:
        import torch
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = getattr(lib, name)(*args)
        b.record()
        TRACE.setdefault(name, []).append((a, b))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise EavError(f"{name} failed ({rc}): {lib.eav_last_error().decode()}")


def plain(name, *args):
    return getattr(load(), name)(*args)
