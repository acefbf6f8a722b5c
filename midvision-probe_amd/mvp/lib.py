"""ctypes binding of libmvp_hip.so (C ABI declared in include/mvp_hip.h).

The product path has NO fallback: if the HIP library is missing or an op returns an error
code, a Python exception is raised.  PyTorch is used only for device memory and streams;
every entry point receives raw device pointers + the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVP_LIB: diagnostic override (A/B builds of the same ABI, e.g. tools/tn_bench.py); the product default is the in-tree library
LIB_PATH = os.environ.get("MVP_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libmvp_hip.so")

PREC_BF16 = 1
PREC_BF16X3 = 3
PREC_F16X2 = 2  # two fp16 products per GEMM contraction over compensated fp16 pairs (ops.split_f16_comp / ops.f16x2_weight): mvp_hip.h
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
RESIZE_NEAREST, RESIZE_BILINEAR, RESIZE_BICUBIC = 0, 1, 2
PAIR_SEPARATE, PAIR_A_ILV32, PAIR_W_ILV32 = 0, 1, 2  # mvp_gemm_args.pair_layout (bit flags)
BN_RUNNING_MAX = 8  # MVP_BN_RUNNING_MAX: modules per mvp_bn_running_update_n launch
TILES_SHARED, TILES_NO_PP, TILES_NO_UNI = 1, 2, 4

_vp = C.c_void_p
_i = C.c_int
_f = C.c_float
_i64 = C.c_int64


class MvpError(RuntimeError):
    pass


class Info(C.Structure):
    _fields_ = [("abi_version", _i), ("device_count", _i), ("gfx950", _i), ("cu_count", _i), ("arch", C.c_char * 64)]


class SplitArgs(C.Structure):
    _fields_ = [("src", _vp), ("hi", _vp), ("lo", _vp), ("n", _i64)]


class PatchGatherArgs(C.Structure):
    _fields_ = [("images", _vp), ("out_hi", _vp), ("out_lo", _vp), ("B", _i), ("C", _i), ("H", _i), ("W", _i),
                ("P", _i), ("gh", _i), ("gw", _i), ("pad_top", _i), ("pad_left", _i)]


class GemmArgs(C.Structure):
    _fields_ = [("a_hi", _vp), ("a_lo", _vp), ("w_hi", _vp), ("w_lo", _vp), ("bias", _vp), ("residual", _vp),
                ("out_f32", _vp), ("out_hi", _vp), ("out_lo", _vp), ("M", _i), ("N", _i), ("K", _i),
                ("lda", _i), ("ldw", _i), ("ldr", _i), ("ldo", _i), ("ldob", _i), ("act", _i), ("precision", _i),
                ("row_group", _i), ("row_group_stride", _i), ("row_group_off", _i), ("res_row_mod", _i),
                ("conv", _i), ("cH", _i), ("cW", _i), ("cC", _i), ("cHo", _i), ("cWo", _i), ("ckh", _i), ("ckw", _i),
                ("cstride", _i), ("cpad", _i), ("cup", _i), ("zero_page", _vp), ("relu_mask", _vp), ("out_mask", _vp), ("ldm", _i), ("mask_mode", _i),
                ("residual2", _vp), ("act_after_res", _i), ("splitk", _i), ("splitk_ws", _vp), ("splitk_ws_bytes", _i64),
                ("residual_hi", _vp), ("residual_lo", _vp), ("tile_policy", _i), ("pair_layout", _i), ("out_pair_layout", _i),
                ("out_f16_col0", _i)]


class LayerNormArgs(C.Structure):
    _fields_ = [("x", _vp), ("gamma", _vp), ("beta", _vp), ("out_hi", _vp), ("out_lo", _vp), ("out_f32", _vp),
                ("M", _i), ("C", _i), ("eps", _f), ("out_layout", _i), ("out_f16", _i)]


class AttentionArgs(C.Structure):
    _fields_ = [("qkv_hi", _vp), ("qkv_lo", _vp), ("out_hi", _vp), ("out_lo", _vp), ("B", _i), ("N", _i), ("H", _i),
                ("ld_qkv", _i), ("ld_out", _i), ("scale", _f), ("precision", _i), ("out_layout", _i), ("v_format", _i), ("out_f16", _i)]


class ClsRowsArgs(C.Structure):
    _fields_ = [("cls", _vp), ("pos0", _vp), ("x", _vp), ("B", _i), ("N", _i), ("C", _i)]


class BnTokensArgs(C.Structure):
    _fields_ = [("x", _vp), ("gamma", _vp), ("beta", _vp), ("running_mean", _vp), ("running_var", _vp), ("stats", _vp),
                ("nchw", _vp), ("tok_hi", _vp), ("tok_lo", _vp), ("ld_tok", _i), ("col_off", _i),
                ("tokT_hi", _vp), ("tokT_lo", _vp), ("ldT", _i),
                ("workspace", _vp), ("workspace_bytes", _i64),
                ("B", _i), ("N", _i), ("C", _i), ("hw", _i), ("eps", _f), ("momentum", _f), ("mode", _i), ("cls_out", _vp), ("num_batches_tracked", _vp),
                ("defer_running", _i), ("groups", _i), ("stats_gstride", _i64), ("nchw_gstride", _i64), ("tok_gstride", _i64), ("cls_gstride", _i64)]


class BnRunningUpdateArgs(C.Structure):
    _fields_ = [("stats", _vp), ("running_mean", _vp), ("running_var", _vp), ("num_batches_tracked", _vp), ("C", _i), ("momentum", _f)]


class PackNchwArgs(C.Structure):
    _fields_ = [("nchw", _vp), ("tok_hi", _vp), ("tok_lo", _vp), ("ld_tok", _i), ("col_off", _i),
                ("tokT_hi", _vp), ("tokT_lo", _vp), ("ldT", _i), ("B", _i), ("C", _i), ("hw", _i)]


class ResizeArgs(C.Structure):
    _fields_ = [("src", _vp), ("dst", _vp), ("planes", _i), ("Hi", _i), ("Wi", _i), ("Ho", _i), ("Wo", _i),
                ("mode", _i), ("align_corners", _i), ("channels_last", _i), ("C", _i), ("scale_h", _f), ("scale_w", _f)]


class DepthPredictArgs(C.Structure):
    _fields_ = [("logits", _vp), ("depth", _vp), ("inv_sum", _vp), ("grad_depth", _vp), ("grad_logits", _vp),
                ("P", _i64), ("K", _i), ("min_depth", _f), ("max_depth", _f), ("kind", _i)]


class DepthLossArgs(C.Structure):
    _fields_ = [("pred", _vp), ("target", _vp), ("loss", _vp), ("grad_pred", _vp), ("workspace", _vp),
                ("workspace_bytes", _i64), ("B", _i), ("HW", _i64), ("w_sig", _f), ("w_grad", _f), ("max_depth", _f),
                ("eps", _f), ("sigma", _f)]


class AngularLossArgs(C.Structure):
    _fields_ = [("pred", _vp), ("gt", _vp), ("mask", _vp), ("loss", _vp), ("grad_pred", _vp), ("workspace", _vp),
                ("workspace_bytes", _i64), ("B", _i), ("Cp", _i), ("HW", _i64), ("eps", _f)]


class ColsumArgs(C.Structure):
    _fields_ = [("x", _vp), ("out", _vp), ("M", _i), ("N", _i), ("ld", _i), ("accumulate", _i), ("workspace", _vp), ("workspace_bytes", _i64)]


class AdamWArgs(C.Structure):
    _fields_ = [("param", _vp), ("grad", _vp), ("exp_avg", _vp), ("exp_avg_sq", _vp), ("hyper", _vp), ("n", _i64),
                ("beta1", _f), ("beta2", _f), ("eps", _f), ("weight_decay", _f), ("grad_scale", _f),
                ("lr", _f), ("bias_c1", _f), ("bias_c2", _f)]


class CorrArgmaxArgs(C.Structure):
    _fields_ = [("src_feat", _vp), ("tgt_feat", _vp), ("kp_xy", _vp), ("out_xy", _vp), ("out_val", _vp),
                ("workspace", _vp), ("workspace_bytes", _i64), ("C", _i), ("h", _i), ("w", _i), ("K", _i), ("heat_out", _vp)]


class Argmax2dArgs(C.Structure):
    _fields_ = [("x", _vp), ("out_xy", _vp), ("K", _i), ("h", _i), ("w", _i), ("max_value", _i)]


class ScaleShiftArgs(C.Structure):
    _fields_ = [("x", _vp), ("scale_shift", _vp), ("grad_out", _vp), ("out", _vp), ("B", _i), ("HW", _i64), ("lo", _f), ("hi", _f),
                ("clamp", _i), ("backward", _i)]


class MetricsBreakdownArgs(C.Structure):
    _fields_ = [("pred", _vp), ("gt", _vp), ("seg", _vp), ("scale_shift", _vp), ("level_sums", _vp), ("seg_sums", _vp),
                ("workspace", _vp), ("workspace_bytes", _i64), ("B", _i), ("H", _i), ("W", _i), ("Cp", _i), ("num_levels", _i), ("num_ids", _i),
                ("t1", _f), ("t2", _f), ("t3", _f)]


class ConvWeightPackArgs(C.Structure):
    _fields_ = [("w", _vp), ("out_hi", _vp), ("out_lo", _vp), ("Cout", _i), ("Cin", _i), ("kh", _i), ("kw", _i), ("mode", _i)]


class UpsampleClArgs(C.Structure):
    _fields_ = [("src", _vp), ("dst_f32", _vp), ("dst_hi", _vp), ("dst_lo", _vp), ("B", _i), ("H", _i), ("W", _i), ("C", _i),
                ("f", _i), ("backward", _i)]


class UpconvBoxsumArgs(C.Structure):
    _fields_ = [("g", _vp), ("out_hi", _vp), ("out_lo", _vp), ("B", _i), ("H", _i), ("W", _i), ("C", _i), ("f", _i)]


class UpconvGatherArgs(C.Structure):
    _fields_ = [("t", _vp), ("bias", _vp), ("out_f32", _vp), ("out_hi", _vp), ("out_lo", _vp), ("out_mask", _vp),
                ("B", _i), ("H", _i), ("W", _i), ("C", _i), ("f", _i), ("act", _i)]


class DepthMetricsArgs(C.Structure):
    _fields_ = [("pred", _vp), ("gt", _vp), ("out", _vp), ("scale_shift", _vp), ("workspace", _vp), ("workspace_bytes", _i64),
                ("B", _i), ("HW", _i64), ("scale_invariant", _i)]


class SnormMetricsArgs(C.Structure):
    _fields_ = [("pred", _vp), ("gt", _vp), ("out", _vp), ("workspace", _vp), ("workspace_bytes", _i64), ("B", _i), ("Cp", _i),
                ("HW", _i64), ("t1", _f), ("t2", _f), ("t3", _f)]


class LinearBinsArgs(C.Structure):
    _fields_ = [("l0", _vp), ("depth", _vp), ("inv_sum", _vp), ("gate", _vp), ("grad_depth", _vp), ("grad_l0", _vp),
                ("B", _i), ("h", _i), ("w", _i), ("K", _i), ("f", _i), ("min_depth", _f), ("max_depth", _f)]


class Im2colArgs(C.Structure):
    _fields_ = [("src", _vp), ("out_hi", _vp), ("out_lo", _vp), ("B", _i), ("C", _i), ("H", _i), ("W", _i), ("Ho", _i), ("Wo", _i),
                ("kh", _i), ("kw", _i), ("stride", _i), ("pad", _i), ("ldk", _i)]


class MaxpoolClArgs(C.Structure):
    _fields_ = [("src", _vp), ("dst_f32", _vp), ("dst_hi", _vp), ("dst_lo", _vp), ("B", _i), ("H", _i), ("W", _i), ("C", _i),
                ("Ho", _i), ("Wo", _i), ("k", _i), ("stride", _i), ("pad", _i)]


class StemArgs(C.Structure):
    _fields_ = [("images", _vp), ("w_hi", _vp), ("w_lo", _vp), ("bias", _vp), ("out_f32", _vp), ("out_hi", _vp), ("out_lo", _vp),
                ("B", _i), ("H", _i), ("W", _i), ("precision", _i)]


class MaskSplitArgs(C.Structure):
    _fields_ = [("src", _vp), ("mask", _vp), ("dst_f32", _vp), ("dst_hi", _vp), ("dst_lo", _vp), ("M", _i64), ("N", _i),
                ("lds", _i), ("ldm", _i), ("ldo", _i), ("relu_mask_out", _vp)]


class GemmTnArgs(C.Structure):
    _fields_ = [("g_hi", _vp), ("g_lo", _vp), ("x_hi", _vp), ("x_lo", _vp), ("partial", _vp), ("dw", _vp), ("zero_page", _vp),
                ("M", _i64), ("Cout", _i), ("Cin", _i), ("ldg", _i), ("ldx", _i), ("H", _i), ("W", _i), ("Ho", _i), ("Wo", _i),
                ("kh", _i), ("kw", _i), ("stride", _i), ("pad", _i), ("up", _i), ("splits", _i), ("accumulate", _i), ("precision", _i)]


# every exported symbol of include/mvp_hip.h: name -> args struct (None = special signature)
SYMBOLS = {
    "mvp_get_info": None,
    "mvp_strerror": None,
    "mvp_sizeof": None,
    "mvp_split_bf16": SplitArgs,
    "mvp_patch_gather": PatchGatherArgs,
    "mvp_gemm_bias_act_res": GemmArgs,
    "mvp_gemm_splitk_workspace_bytes": None,
    "mvp_gemm_streamk_workspace_bytes": None,
    "mvp_gemm_streamk": GemmArgs,
    "mvp_layernorm_fwd": LayerNormArgs,
    "mvp_attention_fwd": AttentionArgs,
    "mvp_cls_rows": ClsRowsArgs,
    "mvp_bn_tokens_workspace_bytes": None,
    "mvp_bn_tokens_to_nchw_fwd": BnTokensArgs,
    "mvp_bn_running_update": BnRunningUpdateArgs,
    "mvp_bn_running_update_n": None,
    "mvp_pack_nchw_tokens": PackNchwArgs,
    "mvp_resize_fwd": ResizeArgs,
    "mvp_resize_bwd": ResizeArgs,
    "mvp_resize_aa_fwd": ResizeArgs,
    "mvp_depth_predict_fwd": DepthPredictArgs,
    "mvp_depth_predict_bwd": DepthPredictArgs,
    "mvp_depth_loss_workspace_bytes": None,
    "mvp_depth_loss_fwd_bwd": DepthLossArgs,
    "mvp_angular_loss_fwd_bwd": AngularLossArgs,
    "mvp_colsum_workspace_bytes": None,
    "mvp_colsum": ColsumArgs,
    "mvp_adamw_step": AdamWArgs,
    "mvp_corr_argmax": CorrArgmaxArgs,
    "mvp_corr_workspace_bytes": None,
    "mvp_argmax_2d": Argmax2dArgs,
    "mvp_scale_shift": ScaleShiftArgs,
    "mvp_metrics_breakdown_workspace_bytes": None,
    "mvp_metrics_breakdown": MetricsBreakdownArgs,
    "mvp_conv_weight_pack": ConvWeightPackArgs,
    "mvp_upsample_nearest_cl": UpsampleClArgs,
    "mvp_upconv3_grad_boxsum": UpconvBoxsumArgs,
    "mvp_upconv3_fwd_gather": UpconvGatherArgs,
    "mvp_mask_split": MaskSplitArgs,
    "mvp_metrics_workspace_bytes": None,
    "mvp_depth_metrics": DepthMetricsArgs,
    "mvp_snorm_metrics": SnormMetricsArgs,
    "mvp_linear_bins_fwd": LinearBinsArgs,
    "mvp_linear_bins_bwd": LinearBinsArgs,
    "mvp_im2col_nchw": Im2colArgs,
    "mvp_maxpool_cl": MaxpoolClArgs,
    "mvp_stem7x7_pool": StemArgs,
    "mvp_gemm_pp": GemmArgs,
    "mvp_gemm_tn_workspace_bytes": None,
    "mvp_gemm_tn_conv": GemmTnArgs,
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libmvp_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MvpError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C midvision-probe_amd/csrc`). There is no CPU fallback for the product path."
        )
    lib = C.CDLL(LIB_PATH)
    for name, st in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        if st is not None:
            fn.argtypes = [C.POINTER(st), _vp]
            fn.restype = _i
    lib.mvp_bn_running_update_n.argtypes = [C.POINTER(BnRunningUpdateArgs), _i, _vp]
    lib.mvp_bn_running_update_n.restype = _i
    lib.mvp_get_info.argtypes = [C.POINTER(Info)]
    lib.mvp_get_info.restype = _i
    lib.mvp_strerror.argtypes = [_i]
    lib.mvp_strerror.restype = C.c_char_p
    lib.mvp_corr_workspace_bytes.argtypes = [_i, _i, _i, _i]
    lib.mvp_corr_workspace_bytes.restype = _i64
    lib.mvp_sizeof.argtypes = [C.c_char_p]
    lib.mvp_sizeof.restype = _i
    lib.mvp_bn_tokens_workspace_bytes.argtypes = [_i, _i]
    lib.mvp_bn_tokens_workspace_bytes.restype = _i64
    lib.mvp_colsum_workspace_bytes.argtypes = [_i, _i]
    lib.mvp_colsum_workspace_bytes.restype = _i64
    lib.mvp_metrics_breakdown_workspace_bytes.argtypes = [_i, _i, _i]
    lib.mvp_metrics_breakdown_workspace_bytes.restype = _i64
    lib.mvp_metrics_workspace_bytes.argtypes = [_i]
    lib.mvp_metrics_workspace_bytes.restype = _i64
    lib.mvp_gemm_streamk_workspace_bytes.argtypes = []
    lib.mvp_gemm_streamk_workspace_bytes.restype = _i64
    lib.mvp_gemm_splitk_workspace_bytes.argtypes = [_i, _i, _i]
    lib.mvp_gemm_splitk_workspace_bytes.restype = _i64
    lib.mvp_gemm_tn_workspace_bytes.argtypes = [_i, _i, _i, _i, _i]
    lib.mvp_gemm_tn_workspace_bytes.restype = _i64
    lib.mvp_depth_loss_workspace_bytes.argtypes = [_i, _i64]
    lib.mvp_depth_loss_workspace_bytes.restype = _i64
    _lib = lib
    return lib


def info() -> Info:
    out = Info()
    check(load().mvp_get_info(C.byref(out)), "mvp_get_info")
    return out


def check(code: int, what: str) -> None:
    if code != 0:
        raise MvpError(f"{what} failed: {load().mvp_strerror(code).decode()} (code {code})")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr() -> int:
    """Handle of the calling thread's current HIP stream on the current device.  Every launch asks for it: the raw accessors cost
    ~0.3 us against ~8 us for torch.cuda.current_stream().cuda_stream (a tenth of a probe step's host time, tools/micro/host_profile.py)."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MvpError("expected a device tensor (the HIP path has no CPU fallback)")
    return t.data_ptr()


# Diagnostic hook (tools/micro/chain_timeline.py): when a list is installed, EVERY entry-point call is bracketed by HIP events recorded on
# the stream it is launched on -> (name, stream, start event, end event).  None in production.
TRACE_CALLS = None


def call(name: str, args: C.Structure) -> None:
    if TRACE_CALLS is None:
        check(getattr(load(), name)(C.byref(args), stream_ptr()), name)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(getattr(load(), name)(C.byref(args), stream_ptr()), name)
    e1.record()
    TRACE_CALLS.append((name, stream_ptr(), e0, e1))
