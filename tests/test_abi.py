"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/mvp_hip.h declares; argument validation returns error codes (no compute, no GPU)."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def _declared():
    src = open(os.path.join(REPO, "include", "mvp_hip.h")).read()
    return sorted(set(re.findall(r"^\s*(?:int|int64_t|const char\*)\s+(mvp_\w+)\s*\(", src, flags=re.M)))


def test_header_symbols_are_exported_and_bound():
    from mvp import lib

    declared = _declared()
    assert len(declared) >= 20
    assert set(declared) == set(lib.SYMBOLS), set(declared) ^ set(lib.SYMBOLS)
    so = lib.load()
    for name in declared:
        assert hasattr(so, name), name


def test_info_and_strerror():
    from mvp import lib

    inf = lib.info()
    assert inf.abi_version == 7
    assert lib.load().mvp_strerror(-1).decode().startswith("invalid argument")


def test_bad_arguments_return_einval():
    """Host-side shape checks run before any launch, so they are testable without a GPU."""
    from mvp import lib

    so = lib.load()
    a = lib.GemmArgs()  # all NULL / zero
    assert so.mvp_gemm_bias_act_res(ctypes.byref(a), None) == -1
    a = lib.GemmArgs(a_hi=16, w_hi=16, out_f32=16, M=4, N=4, K=48, lda=48, ldw=48, precision=1)
    assert so.mvp_gemm_bias_act_res(ctypes.byref(a), None) == -1  # K % 64 != 0
    b = lib.AttentionArgs(qkv_hi=16, out_hi=16, B=1, N=4, H=1, ld_qkv=100, ld_out=64, precision=1)
    assert so.mvp_attention_fwd(ctypes.byref(b), None) == -1  # ld_qkv < 3*H*64
    c = lib.LayerNormArgs(x=16, gamma=16, beta=16, out_hi=16, M=1, C=4098)
    assert so.mvp_layernorm_fwd(ctypes.byref(c), None) == -1
    assert so.mvp_bn_tokens_workspace_bytes(3152, 768) == (394 * 768 * 3 + 2 * 768) * 4  # 8-row slabs
    assert so.mvp_colsum_workspace_bytes(3136, 256) == 49 * 256 * 4


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mvp import lib

    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lib.MvpError, match="no CPU fallback"):
        lib.load()


def test_ctypes_structs_match_the_compiled_header():
    """Every ctypes mirror in mvp/lib.py has the size the C compiler gave the struct of include/mvp_hip.h
    (mvp_sizeof): catches field-order / padding drift between the header and the binding."""
    from mvp import lib

    so = lib.load()
    names = {
        "mvp_info_t": lib.Info, "mvp_split_bf16_args": lib.SplitArgs, "mvp_patch_gather_args": lib.PatchGatherArgs,
        "mvp_gemm_args": lib.GemmArgs, "mvp_layernorm_args": lib.LayerNormArgs, "mvp_attention_args": lib.AttentionArgs,
        "mvp_cls_rows_args": lib.ClsRowsArgs, "mvp_bn_tokens_args": lib.BnTokensArgs, "mvp_pack_nchw_args": lib.PackNchwArgs,
        "mvp_resize_args": lib.ResizeArgs, "mvp_depth_predict_args": lib.DepthPredictArgs, "mvp_depth_loss_args": lib.DepthLossArgs,
        "mvp_angular_loss_args": lib.AngularLossArgs, "mvp_colsum_args": lib.ColsumArgs, "mvp_adamw_args": lib.AdamWArgs,
        "mvp_corr_argmax_args": lib.CorrArgmaxArgs, "mvp_conv_weight_pack_args": lib.ConvWeightPackArgs,
        "mvp_upsample_cl_args": lib.UpsampleClArgs, "mvp_gemm_tn_args": lib.GemmTnArgs, "mvp_depth_metrics_args": lib.DepthMetricsArgs,
        "mvp_snorm_metrics_args": lib.SnormMetricsArgs, "mvp_linear_bins_args": lib.LinearBinsArgs, "mvp_im2col_args": lib.Im2colArgs,
        "mvp_maxpool_cl_args": lib.MaxpoolClArgs, "mvp_mask_split_args": lib.MaskSplitArgs,
        "mvp_metrics_breakdown_args": lib.MetricsBreakdownArgs, "mvp_argmax_2d_args": lib.Argmax2dArgs, "mvp_scale_shift_args": lib.ScaleShiftArgs, "mvp_stem_args": lib.StemArgs,
        "mvp_bn_running_update_args": lib.BnRunningUpdateArgs, "mvp_upconv_boxsum_args": lib.UpconvBoxsumArgs, "mvp_upconv_gather_args": lib.UpconvGatherArgs,
    }
    import re
    header = open(os.path.join(os.path.dirname(__file__), "..", "include", "mvp_hip.h")).read()
    declared = set(re.findall(r"^\} (mvp_\w+);", header, flags=re.M))
    assert declared == set(names), declared ^ set(names)
    for cname, cls in names.items():
        assert so.mvp_sizeof(cname.encode()) == ctypes.sizeof(cls), (cname, so.mvp_sizeof(cname.encode()), ctypes.sizeof(cls))
    assert so.mvp_sizeof(b"nope") == -1


def test_integration_md_binding_matches_the_header():
    """INTEGRATION.md §3 shows the ctypes struct a maintainer would copy.  Execute exactly that snippet's class definition and hold its
    size to the compiled header (mvp_sizeof) and its field names / order to mvp/lib.py's mirror: a field added to mvp_gemm_args without
    updating the document fails here (boundary doc drift)."""
    import ctypes as C

    from mvp import lib

    md = open(os.path.join(REPO, "INTEGRATION.md")).read()
    m = re.search(r"class GemmArgs\(C\.Structure\):.*?\n(    _fields_ = .*?\])\s*\n\s*#", md, flags=re.S)
    assert m, "INTEGRATION.md: the GemmArgs binding snippet was not found"
    ns = {"C": C, "_vp": C.c_void_p, "_i": C.c_int}
    exec("class GemmArgs(C.Structure):\n" + m.group(1), ns)
    doc = ns["GemmArgs"]
    assert [f[0] for f in doc._fields_] == [f[0] for f in lib.GemmArgs._fields_]
    assert C.sizeof(doc) == lib.load().mvp_sizeof(b"mvp_gemm_args") == C.sizeof(lib.GemmArgs)
