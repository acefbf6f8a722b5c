"""Tensor-level wrappers over the C ABI (one function per entry point).

A "pair" is a tuple ``(hi, lo)`` of torch.bfloat16 tensors with value hi + lo; ``lo`` is
None in MVP_PREC_BF16 mode.  All tensors must live on the current HIP device; outputs are
allocated by the caller (or by these helpers with torch.empty — plumbing only).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import lib, pipeline
from .lib import PREC_BF16, PREC_BF16X3

Pair = Tuple[torch.Tensor, Optional[torch.Tensor]]


class IlvPair:
    """A bf16 pair stored as ONE array, hi | lo interleaved per 32 columns (MVP_PAIR_A_ILV32, include/mvp_hip.h): ``t`` is
    [rows, 2 * cols] bf16.  LayerNorm, attention and the GEMM epilogue write it, the large-M GEMM kernel reads it as its A operand
    (a 32-deep k-step of a row is then one whole 128-byte line)."""

    def __init__(self, rows: int, cols: int, device):
        if cols % 32:
            raise lib.MvpError("an interleaved pair needs cols % 32 == 0")
        self.rows, self.cols = rows, cols
        self.t = torch.empty(rows, 2 * cols, dtype=torch.bfloat16, device=device)

    def separate(self) -> Pair:
        """The same values as two [rows, cols] arrays (tests)."""
        v = self.t.view(self.rows, self.cols // 32, 2, 32)
        return v[:, :, 0].reshape(self.rows, self.cols).contiguous(), v[:, :, 1].reshape(self.rows, self.cols).contiguous()

# Optional per-launch timing hook used by bench.py's roofline leg: when a list is installed,
# every GEMM / attention launch is bracketed by HIP events recorded on the launch stream.
_TRACE = None


def set_trace(lst) -> None:
    global _TRACE
    _TRACE = lst


def _traced(kind: str, tile: str, precision: int, work: float, fn) -> None:
    """Run ``fn`` bracketed by HIP events on the launch stream when bench.py has installed a trace list
    (``work`` = algorithmic flops for MFMA-bound kernels, algorithmic HBM bytes for memory-bound ones)."""
    if _TRACE is None:
        fn()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    _TRACE.append((kind, tile, precision, work, e0, e1))


def gemm_tile(M: int, N: int, K: int = 0, precision: int = PREC_BF16X3, splitk: int = 1, tile_policy: int = 0) -> str:
    """Mirror of the tile choice in csrc/gemm.hip -> template arguments BM,BN,BK,SPLIT,NSTAGE."""
    if splitk > 1:
        bn = 128 if N >= 1024 else 64
        return f"128, {bn}, 64, 3, 1, splitk" if precision == PREC_BF16X3 else f"128, {bn}, 64, 1, 2, splitk"
    if precision in (PREC_BF16X3, lib.PREC_F16X2) and K % 32 == 0 and K >= 64 and not (N <= 256 and K >= 2048) and os.environ.get("MVP_GEMM_PP", "") != "0":
        t256 = ((M + 255) // 256) * ((N + 255) // 256)  # pp_takes() of csrc/gemm.hip: the large-M ping-pong kernel
        rounds = (t256 + 255) // 256
        narrow = ((N + 255) // 256) * 256 * 7 > N * 8  # more than 1/8 of the tile grid's columns would be padding
        if os.environ.get("MVP_GEMM_PP") == "1" or not narrow and (t256 >= 96 if tile_policy == 1 else (t256 >= 128 if K >= 2048 else t256 >= 200 and (t256 * 5 >= rounds * 1024 or t256 >= 1024))):
            return f"pp 256, 256, 32, {2 if precision == lib.PREC_F16X2 else 3}"
    t128 = ((M + 127) // 128) * ((N + 127) // 128)
    if precision == lib.PREC_F16X2:  # the reduced rule of csrc/gemm.hip for the two-product mode
        if N >= 1024:
            return "128, 128, 64, 2, 1"
        return "64, 64, 64, 2, 1" if ((M + 63) // 64) * ((N + 63) // 64) <= 1280 else "128, 64, 64, 2, 1"
    if precision == PREC_BF16X3:
        if N <= 256 and K >= 2048:
            return "128, 64, 64, 3, 1" if M >= 8192 else "64, 64, 64, 3, 2"
        if tile_policy == 1:  # MVP_TILES_SHARED
            return "128, 128, 64, 3, 1"
        if N >= 1024:
            return "128, 128, 64, 3, 1" if (t128 <= 512 or t128 >= 1536) else "64, 128, 64, 3, 1"
        if K >= 4096 and t128 >= 300:
            return "128, 128, 64, 3, 1"
        t64 = ((M + 63) // 64) * ((N + 63) // 64)
        return "64, 64, 64, 3, 1" if t64 <= 1280 else "128, 64, 64, 3, 1"
    return "128, 128, 64, 1, 2" if t128 >= 400 else "64, 64, 64, 1, 2"


def _chk(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype or not t.is_cuda or not t.is_contiguous():
        raise lib.MvpError(f"{name}: expected contiguous {dtype} device tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")


def empty_pair(shape, precision: int, device) -> Pair:
    hi = torch.empty(shape, dtype=torch.bfloat16, device=device)
    lo = torch.empty(shape, dtype=torch.bfloat16, device=device) if precision == PREC_BF16X3 else None
    return hi, lo


def zeros_pair(shape, precision: int, device) -> Pair:
    hi = torch.zeros(shape, dtype=torch.bfloat16, device=device)
    lo = torch.zeros(shape, dtype=torch.bfloat16, device=device) if precision == PREC_BF16X3 else None
    return hi, lo


def split_bf16(src: torch.Tensor, precision: int = PREC_BF16X3) -> Pair:
    src = src.contiguous()
    _chk(src, torch.float32, "split_bf16.src")
    hi, lo = empty_pair(src.shape, precision, src.device)
    a = lib.SplitArgs(lib.ptr(src), lib.ptr(hi), lib.ptr(lo), src.numel())
    lib.call("mvp_split_bf16", a)
    return hi, lo


F16X2_S = 2.0 ** -6  # the share of the hi product that PREC_F16X2 moves into the lo product (include/mvp_hip.h, MVP_PREC_F16X2)


def f16x2_weight(w: torch.Tensor) -> Pair:
    """The weight operand of a PREC_F16X2 GEMM (include/mvp_hip.h): hi = fp16((1 - s) w), lo = fp16((w + d / s) / 8) with
    d = (1 - s) w - hi — fp16 bits in bf16-typed arrays, like every pair half of this library.  fp64 arithmetic: one rounding each."""
    w = w.detach().double().contiguous()
    t = (1.0 - F16X2_S) * w
    hi = t.clamp(-65504.0, 65504.0).to(torch.float16)
    lo = ((w + (t - hi.double()) / F16X2_S) / 8.0).clamp(-65504.0, 65504.0).to(torch.float16)
    return hi.view(torch.bfloat16).contiguous(), lo.view(torch.bfloat16).contiguous()


def split_f16_comp(src: torch.Tensor) -> Pair:
    """fp32 -> the activation pair of a PREC_F16X2 GEMM: hi = fp16(x), lo = fp16(8 (x - hi) + hi / 8), both saturating at +-65504
    (bits in bf16-typed arrays).  (torch ops: tests and one-off conversions; the kernels write this form themselves.)"""
    x = src.float()
    hi = x.clamp(-65504.0, 65504.0).half()
    h = hi.float()
    lo = ((x - h) * 8.0 + h * 0.125).clamp(-65504.0, 65504.0).half()
    return hi.view(torch.bfloat16).contiguous(), lo.view(torch.bfloat16).contiguous()


def split_f16_bf16(src: torch.Tensor) -> Pair:
    """fp32 -> hi = fp16(x) (bits in a bf16-typed array, saturating), lo = bf16(x - hi): the form of the V third under
    ``gemm(..., f16_col0=2 * H * 64)``.  (torch ops: tests.)"""
    x = src.float()
    hi = x.clamp(-65504.0, 65504.0).half()
    return hi.view(torch.bfloat16).contiguous(), (x - hi.float()).bfloat16().contiguous()


def interleave_pair(pair: Pair) -> Optional[torch.Tensor]:
    """(hi, lo) [R, K] -> ONE [R, 2K] bf16 array, hi | lo interleaved per 32-deep k block (MVP_PAIR_*_ILV32, include/mvp_hip.h): the
    operand layout in which a 32-deep k-step of a row is one whole 128-byte line.  Used for the frozen weights (built once at load; the
    large-M GEMM kernel reads them 2-3.5 % faster than the separate arrays).  None when K % 32 != 0 or the pair has no lo part."""
    hi, lo = pair
    if lo is None or hi.dim() != 2 or hi.shape[1] % 32:
        return None
    R, K = hi.shape
    return torch.stack((hi.view(R, K // 32, 32), lo.view(R, K // 32, 32)), dim=2).reshape(R, 2 * K).contiguous()


def patch_gather(images: torch.Tensor, out: Pair, P: int, gh: int, gw: int, pad_top: int, pad_left: int) -> None:
    _chk(images, torch.float32, "patch_gather.images")
    B, Cc, H, W = images.shape
    a = lib.PatchGatherArgs(lib.ptr(images), lib.ptr(out[0]), lib.ptr(out[1]), B, Cc, H, W, P, gh, gw, pad_top, pad_left)
    lib.call("mvp_patch_gather", a)


_SPLITK_WS = {}  # (device, stream) -> zero-initialised workspace (tile counters reset themselves)


def splitk_auto(M: int, N: int, K: int) -> int:
    """Split-K factor used when the caller does not give one.  Measured on MI355X (tools/splitk_bench.py and in-step,
    bf16x3): the N = 768 projections do NOT profit (proj 27.1 us unsplit vs 33.6 at S=2; fc2 71.8 vs 77.3): they are
    bound by the L2->LDS operand path, not by tile imbalance.  The one shape that does is the probe head at small M
    (skinny N <= 256, long K, <= 128 tiles): its A operand was just written by the tap kernels and comes from HBM /
    Infinity Cache, and 4x more workgroups pull it 4x wider: 52.5 -> 40.4 us inside the step at B=16.
    (The intra-launch partial-tile hand-over was suspected when pipelined trajectories stopped reproducing the serial ones; the cause
    was elsewhere — csrc/Makefile, packed fp32 — and 1500 trajectories with this rule beside other kernel chains are bit-exact,
    tools/micro/race_hunt2.py VAR=splitk.  The hand-over carries an agent-scope release since.)"""
    if N <= 256 and K >= 2048 and ((M + 127) // 128) * ((N + 63) // 64) <= 128:
        return 4
    return 1


_STREAMK_WS = {}  # (device, stream) -> zero-initialised stream-K workspace (tile counters reset themselves)
_STREAMK_MODE = os.environ.get("MVP_STREAMK", "auto")  # "auto" | "0" | "1" (diagnostic override)


def streamk_auto(M: int, N: int, K: int, precision: int) -> bool:
    """Stream-K scheduling (csrc/gemm_sk.hip: 128x128x64 tiles, one workgroup per CU, equal k-iterations per CU) is OPT-IN
    (MVP_STREAMK=1 or streamk=True).  Measured on MI355X (tools/gemm_bench.py --sk, bf16x3, us: tile kernel -> stream-K best
    variant): B=16 qkv 39.8 -> 54.5, proj 20.3 -> 30.7, fc1 50.7 -> 69.1, fc2 61.5 -> 79.5; B=64 qkv 131 -> 170, fc2 169 -> 261.
    One lock-stepped 8-wave workgroup per CU issues its LDS-DMA pieces (each costs the issuing wave 75-185 cycles) from far fewer
    waves than five independent 64x64 workgroups do, and the partial-tile hand-over adds ~15 us per launch."""
    return _STREAMK_MODE == "1"


def _ws_key(device):
    # launches on one stream are ordered, so they may share a workspace; launches on different streams (mvp/pipeline.py) may not.
    # A pipelined forward is keyed by its SLOT, not by the stream it happens to be enqueued on: every slot's hipGraph is captured on
    # one stream and replayed on whichever stream its turn falls on, so a stream-keyed workspace would be baked into several graphs
    # that replay side by side
    if pipeline.pipelined():
        return (device, "slot", pipeline.current_slot())
    return (device, lib.stream_ptr())


def _streamk_workspace(device) -> torch.Tensor:
    key = _ws_key(device)
    ws = _STREAMK_WS.get(key)
    if ws is None:
        ws = _STREAMK_WS[key] = torch.zeros(int(lib.load().mvp_gemm_streamk_workspace_bytes()), dtype=torch.uint8, device=device)
    return ws


def _splitk_workspace(M: int, N: int, S: int, device) -> torch.Tensor:
    need = int(lib.load().mvp_gemm_splitk_workspace_bytes(M, N, S))
    key = _ws_key(device)
    ws = _SPLITK_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _SPLITK_WS[key] = torch.zeros(max(need, 32 << 20), dtype=torch.uint8, device=device)
    return ws


def gemm(a: Pair, w: Pair, M: int, N: int, K: int, *, bias=None, residual=None, out_f32=None, out: Optional[Pair] = None,
         act: int = lib.ACT_NONE, precision: int = PREC_BF16X3, lda=None, ldw=None, ldr=None, ldo=None, ldob=None,
         row_group=0, row_group_stride=0, row_group_off=0, res_row_mod=0, act_after_res=False, out_mask=None, ldm=0,
         splitk: Optional[int] = None, residual_pair: Optional[Pair] = None, streamk: Optional[bool] = None,
         w_ilv: Optional[torch.Tensor] = None, f16_col0: int = 0) -> None:
    """Y = act(A Wᵀ + bias) + residual (see mvp_gemm_bias_act_res).  splitk: None = automatic, 1 = off.
    f16_col0 > 0: columns from there on of the pair output are written as hi = fp16, lo = bf16 (mvp_gemm_args.out_f16_col0: the V
    third of the fused qkv projection for ``attention(..., v_f16=True)``); f16_col0 = -1: EVERY column as the activation operand of a
    PREC_F16X2 GEMM (``split_f16_comp``'s form: fc1 -> fc2); no split-K / stream-K either way.
    residual_pair: the residual as a bf16 pair (hi, lo) instead of / in addition to the fp32 ``residual``.
    streamk: None = automatic (streamk_auto), True / False force the stream-K kernel on / off.
    w_ilv: the same weights as ``interleave_pair(w)``; handed to the large-M kernel when the dispatch rule picks it."""
    a_ilv, o_ilv = isinstance(a, IlvPair), isinstance(out, IlvPair)
    if a_ilv:
        a, lda = (a.t, None), (lda if lda is not None else 2 * K)
    if o_ilv:
        out, ldob = (out.t, None), (ldob if ldob is not None else 2 * N)
    o_hi, o_lo = out if out is not None else (None, None)
    args = lib.GemmArgs(
        lib.ptr(a[0]), lib.ptr(a[1]), lib.ptr(w[0]), lib.ptr(w[1]), lib.ptr(bias), lib.ptr(residual),
        lib.ptr(out_f32), lib.ptr(o_hi), lib.ptr(o_lo), M, N, K,
        lda if lda is not None else K, ldw if ldw is not None else K,
        ldr if ldr is not None else N, ldo if ldo is not None else N, ldob if ldob is not None else N,
        act, precision, row_group, row_group_stride, row_group_off, res_row_mod)
    args.act_after_res = int(act_after_res)
    args.tile_policy = pipeline.tile_policy()
    args.pair_layout = lib.PAIR_A_ILV32 if a_ilv else lib.PAIR_SEPARATE  # (an interleaved A operand always goes to the large-M kernel)
    args.out_pair_layout = lib.PAIR_A_ILV32 if o_ilv else lib.PAIR_SEPARATE
    if f16_col0:
        args.out_f16_col0 = int(f16_col0)
        splitk, streamk = 1, False
    if precision == lib.PREC_F16X2:  # two products per contraction: plain linear GEMMs, no split-K / stream-K (mvp_hip.h)
        splitk, streamk = 1, False
    if out_mask is not None:
        args.out_mask, args.ldm = lib.ptr(out_mask), ldm or N
    if residual_pair is not None:
        args.residual_hi, args.residual_lo = lib.ptr(residual_pair[0]), lib.ptr(residual_pair[1])
    S = 1
    plain = out_mask is None and not act_after_res and residual_pair is None
    if (a_ilv or o_ilv) and (not plain or splitk not in (None, 1) or streamk):
        raise lib.MvpError("interleaved pair operands: plain GEMMs only (no masks, split-K or stream-K)")
    use_sk = plain and not (a_ilv or o_ilv) and splitk is None and (streamk_auto(M, N, K, precision) if streamk is None else bool(streamk))
    if use_sk:
        ws = _streamk_workspace(a[0].device)
        args.splitk, args.splitk_ws, args.splitk_ws_bytes = -1, lib.ptr(ws), ws.numel()
    elif plain and not a_ilv:
        S = splitk_auto(M, N, K) if splitk is None else int(splitk)
    if S > 1:
        ws = _splitk_workspace(M, N, S, a[0].device)
        args.splitk, args.splitk_ws, args.splitk_ws_bytes = S, lib.ptr(ws), ws.numel()
    tile = gemm_tile(M, N, K, precision, S, args.tile_policy) if (w_ilv is not None or _TRACE is not None) and not a_ilv else ""
    if a_ilv:
        tile = f"pp 256, 256, 32, {2 if precision == lib.PREC_F16X2 else 3}"
    if w_ilv is not None and plain and not use_sk and S <= 1 and tile.startswith("pp ") and ldw is None:
        args.w_hi, args.w_lo, args.ldw = lib.ptr(w_ilv), None, 2 * K
        args.pair_layout |= lib.PAIR_W_ILV32
    if _TRACE is None:
        lib.call("mvp_gemm_bias_act_res", args)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.call("mvp_gemm_bias_act_res", args)
    e1.record()
    _TRACE.append(("gemm", "streamk 128, 128, 64" if use_sk else tile, precision, 2.0 * M * N * K, e0, e1))


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out: Pair, M: int, Cdim: int, eps: float,
              out_f32: Optional[torch.Tensor] = None, out_f16: bool = False) -> None:
    """``out_f16``: the pair is written as the activation operand of a PREC_F16X2 GEMM (``split_f16_comp``'s form)."""
    ilv = isinstance(out, IlvPair)
    if ilv:
        out = (out.t, None)
    a = lib.LayerNormArgs(lib.ptr(x), lib.ptr(gamma), lib.ptr(beta), lib.ptr(out[0]), lib.ptr(out[1]), lib.ptr(out_f32), M, Cdim, eps,
                          lib.PAIR_A_ILV32 if ilv else lib.PAIR_SEPARATE, 1 if out_f16 else 0)
    # algorithmic HBM bytes: the fp32 row read once + the bf16 pair (or single bf16) written once
    nb = M * Cdim * (4 + 2 * (2 if (ilv or out[1] is not None) else 1) + (4 if out_f32 is not None else 0))
    _traced("hbm", "layernorm_kernel", 0, float(nb), lambda: lib.call("mvp_layernorm_fwd", a))


def attention(qkv: Pair, out: Pair, B: int, N: int, H: int, scale: float, precision: int, ld_qkv=None, ld_out=None, v_f16: bool = False,
              out_f16: bool = False, qk_f16: bool = False) -> None:
    """``v_f16``: the V third of ``qkv`` holds hi = fp16, lo = bf16 (``gemm(..., f16_col0=2 * H * 64)``); the probabilities are then held
    as one fp16 value (mvp_attention_args.v_format = MVP_ATT_V_F16; bf16x3 only).  ``out_f16``: the output pair leaves as the activation
    operand of a PREC_F16X2 GEMM (``split_f16_comp``'s form).  ``qk_f16`` (with ``v_f16``): Q and K are the compensated fp16 pairs of
    ``gemm(..., f16_col0=-2 * H * 64)`` (activation / weight-side form) and Q.K^T runs two f16 products (MVP_ATT_V_F16_QK_F16)."""
    if qk_f16 and not v_f16:
        raise lib.MvpError("attention: qk_f16 needs v_f16 (mvp_attention_args.v_format = MVP_ATT_V_F16_QK_F16)")
    ilv = isinstance(out, IlvPair)
    if ilv:
        out, ld_out = (out.t, None), (ld_out if ld_out is not None else 2 * H * 64)
    a = lib.AttentionArgs(lib.ptr(qkv[0]), lib.ptr(qkv[1]), lib.ptr(out[0]), lib.ptr(out[1]), B, N, H,
                          ld_qkv if ld_qkv is not None else 3 * H * 64, ld_out if ld_out is not None else H * 64, scale, precision,
                          lib.PAIR_A_ILV32 if ilv else lib.PAIR_SEPARATE, (2 if qk_f16 else 1) if v_f16 else 0, 1 if out_f16 else 0)
    if _TRACE is None:
        lib.call("mvp_attention_fwd", a)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.call("mvp_attention_fwd", a)
    e1.record()
    _TRACE.append(("attention", "4x32q", precision, 4.0 * B * H * N * N * 64, e0, e1))


def cls_rows(cls: torch.Tensor, pos0: torch.Tensor, x: torch.Tensor, B: int, N: int, Cdim: int) -> None:
    lib.call("mvp_cls_rows", lib.ClsRowsArgs(lib.ptr(cls), lib.ptr(pos0), lib.ptr(x), B, N, Cdim))


def bn_tokens_workspace_bytes(M: int, Cdim: int) -> int:
    return int(lib.load().mvp_bn_tokens_workspace_bytes(M, Cdim))


def bn_tokens_to_nchw(x, B, N, Cdim, hw, *, workspace, stats=None, gamma=None, beta=None, running_mean=None, running_var=None,
                      nchw=None, tok: Optional[Pair] = None, ld_tok=0, col_off=0, tokT: Optional[Pair] = None, ldT=0,
                      eps=1e-5, momentum=0.1, mode=0, cls_out=None, num_batches_tracked=None, defer_running=False,
                      groups=1, stats_gstride=0, nchw_gstride=0, tok_gstride=0, cls_gstride=0) -> None:
    """``defer_running`` (mode 0): the running statistics and the step counter are left alone, ``stats`` ([3*C]) also receives the
    unbiased variance, and ``bn_running_update`` applies the momentum update later (forwards in flight on several streams).
    ``groups`` = G > 1: ``x`` holds G stacked batches of B images; every batch is normalised with its own statistics in the same three
    launches, outputs at the given element strides between batches (mvp_bn_tokens_args.groups)."""
    t_hi, t_lo = tok if tok is not None else (None, None)
    tt_hi, tt_lo = tokT if tokT is not None else (None, None)
    a = lib.BnTokensArgs(lib.ptr(x), lib.ptr(gamma), lib.ptr(beta), lib.ptr(running_mean), lib.ptr(running_var), lib.ptr(stats),
                         lib.ptr(nchw), lib.ptr(t_hi), lib.ptr(t_lo), ld_tok, col_off, lib.ptr(tt_hi), lib.ptr(tt_lo), ldT,
                         lib.ptr(workspace), workspace.numel() * workspace.element_size(), B, N, Cdim, hw, eps, momentum, mode, lib.ptr(cls_out),
                         lib.ptr(num_batches_tracked) if mode == 0 else None, int(bool(defer_running) and mode == 0),
                         int(groups), int(stats_gstride), int(nchw_gstride), int(tok_gstride), int(cls_gstride))
    # algorithmic HBM bytes: x read twice in train mode (statistics, apply), NCHW fp32 + token-major pair written
    M = B * N * max(1, int(groups))
    B = B * max(1, int(groups))
    nb = M * Cdim * 4 * (2 if mode == 0 else 1) + B * hw * Cdim * ((4 if nchw is not None else 0) + (4 if tok is not None and tok[1] is not None else (2 if tok is not None else 0)))
    _traced("hbm", "bn_tokens (partial+finalize+apply)", 0, float(nb), lambda: lib.call("mvp_bn_tokens_to_nchw_fwd", a))


def bn_running_update(stats, running_mean, running_var, num_batches_tracked, Cdim, momentum=0.1) -> None:
    """The deferred running-statistics update of a ``bn_tokens_to_nchw(..., defer_running=True)`` call (same bits as the fused form)."""
    lib.call("mvp_bn_running_update", lib.BnRunningUpdateArgs(lib.ptr(stats), lib.ptr(running_mean), lib.ptr(running_var),
                                                               lib.ptr(num_batches_tracked), Cdim, momentum))


def bn_running_update_many(items, momentum=0.1) -> None:
    """``bn_running_update`` for several BatchNorm modules — items of (stats, running_mean, running_var, num_batches_tracked, Cdim) — in
    one launch per lib.BN_RUNNING_MAX of them (mvp_bn_running_update_n; element-wise the same bits)."""
    if len({it[1].data_ptr() for it in items}) != len(items):  # one module tapped twice: its updates are ordered, one launch each
        for st, rm, rv, nbt, Cdim in items:
            bn_running_update(st, rm, rv, nbt, Cdim, momentum)
        return
    for i0 in range(0, len(items), lib.BN_RUNNING_MAX):
        part = items[i0:i0 + lib.BN_RUNNING_MAX]
        arr = (lib.BnRunningUpdateArgs * len(part))(*[
            lib.BnRunningUpdateArgs(lib.ptr(st), lib.ptr(rm), lib.ptr(rv), lib.ptr(nbt), Cdim, momentum) for st, rm, rv, nbt, Cdim in part])
        lib.check(lib.load().mvp_bn_running_update_n(arr, len(part), lib.stream_ptr()), "mvp_bn_running_update_n")


def pack_nchw_tokens(nchw: torch.Tensor, B: int, Cdim: int, hw: int, *, tok: Optional[Pair] = None, ld_tok=0, col_off=0,
                     tokT: Optional[Pair] = None, ldT=0) -> None:
    t_hi, t_lo = tok if tok is not None else (None, None)
    tt_hi, tt_lo = tokT if tokT is not None else (None, None)
    a = lib.PackNchwArgs(lib.ptr(nchw), lib.ptr(t_hi), lib.ptr(t_lo), ld_tok, col_off, lib.ptr(tt_hi), lib.ptr(tt_lo), ldT, B, Cdim, hw)
    lib.call("mvp_pack_nchw_tokens", a)


def resize(src: torch.Tensor, dst: torch.Tensor, planes: int, Hi: int, Wi: int, Ho: int, Wo: int, mode: int, *, align_corners=False,
           channels_last=False, Cdim=0, scale_h=0.0, scale_w=0.0, backward=False) -> None:
    """forward: src [.,Hi,Wi] -> dst [.,Ho,Wo];  backward: src = grad_out [.,Ho,Wo] -> dst = grad_in [.,Hi,Wi]."""
    a = lib.ResizeArgs(lib.ptr(src), lib.ptr(dst), planes, Hi, Wi, Ho, Wo, mode, int(align_corners), int(channels_last), Cdim, scale_h, scale_w)
    lib.call("mvp_resize_bwd" if backward else "mvp_resize_fwd", a)


def depth_predict_fwd(logits, depth, inv_sum, P, K, min_depth, max_depth, kind) -> None:
    a = lib.DepthPredictArgs(lib.ptr(logits), lib.ptr(depth), lib.ptr(inv_sum), None, None, P, K, min_depth, max_depth, kind)
    lib.call("mvp_depth_predict_fwd", a)


def depth_predict_bwd(logits, depth, inv_sum, grad_depth, grad_logits, P, K, min_depth, max_depth, kind) -> None:
    a = lib.DepthPredictArgs(lib.ptr(logits), lib.ptr(depth), lib.ptr(inv_sum), lib.ptr(grad_depth), lib.ptr(grad_logits), P, K, min_depth, max_depth, kind)
    lib.call("mvp_depth_predict_bwd", a)


def depth_loss_workspace_bytes(B: int, HW: int) -> int:
    return int(lib.load().mvp_depth_loss_workspace_bytes(B, HW))


def depth_loss(pred, target, loss, grad_pred, workspace, B, HW, w_sig=10.0, w_grad=0.5, max_depth=10.0, eps=1e-3, sigma=0.85) -> None:
    a = lib.DepthLossArgs(lib.ptr(pred), lib.ptr(target), lib.ptr(loss), lib.ptr(grad_pred), lib.ptr(workspace),
                          workspace.numel() * workspace.element_size(), B, HW, w_sig, w_grad, max_depth, eps, sigma)
    # algorithmic HBM bytes: pred + target read, gradient written (one pass each)
    _traced("hbm", "depth_loss (dlf_stats+dlf_grad)" if B <= 16 else "depth_loss (dl_stats+dl_pairs+dl_finalize+dl_grad)", 0, float(B * HW * 12), lambda: lib.call("mvp_depth_loss_fwd_bwd", a))


def angular_loss(pred, gt, mask_u8, loss, grad_pred, workspace, B, Cp, HW, eps=1e-4) -> None:
    a = lib.AngularLossArgs(lib.ptr(pred), lib.ptr(gt), lib.ptr(mask_u8), lib.ptr(loss), lib.ptr(grad_pred), lib.ptr(workspace),
                            workspace.numel() * workspace.element_size(), B, Cp, HW, eps)
    lib.call("mvp_angular_loss_fwd_bwd", a)


def colsum(x, out, M, N, ld=None, accumulate=False, workspace=None) -> None:
    if workspace is None:
        workspace = torch.empty(int(lib.load().mvp_colsum_workspace_bytes(M, N)) // 4 + 4, dtype=torch.float32, device=x.device)
    lib.call("mvp_colsum", lib.ColsumArgs(lib.ptr(x), lib.ptr(out), M, N, ld if ld is not None else N, int(accumulate),
                                          lib.ptr(workspace), workspace.numel() * workspace.element_size()))


def adamw_step(param, grad, exp_avg, exp_avg_sq, hyper, n, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, grad_scale=1.0,
               lr=0.0, bias_c1=0.0, bias_c2=0.0) -> None:
    """hyper: device tensor {lr, 1-beta1^t, 1-beta2^t} or None (then lr / bias_c1 / bias_c2 go by value)."""
    a = lib.AdamWArgs(lib.ptr(param), lib.ptr(grad), lib.ptr(exp_avg), lib.ptr(exp_avg_sq), lib.ptr(hyper), n, beta1, beta2, eps, weight_decay, grad_scale,
                      lr, bias_c1, bias_c2)
    lib.call("mvp_adamw_step", a)


def corr_argmax(src_feat, tgt_feat, kp_xy, out_xy, out_val, workspace, Cdim, h, w, K, heat_out=None) -> None:
    a = lib.CorrArgmaxArgs(lib.ptr(src_feat), lib.ptr(tgt_feat), lib.ptr(kp_xy), lib.ptr(out_xy), lib.ptr(out_val), lib.ptr(workspace),
                           workspace.numel() * workspace.element_size(), Cdim, h, w, K, lib.ptr(heat_out))
    lib.call("mvp_corr_argmax", a)
