"""Tensor-level wrappers of the convolution ops (channels-last bf16 pairs; see csrc/conv.hip and
the conv mode of csrc/gemm.hip).  Geometry dicts: B, H, W (virtual input dims, i.e. after the
optional nearest upsample by 2**up), C (input channels), Ho, Wo, kh, kw, stride, pad, up."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import lib, ops, pipeline
from .lib import PREC_BF16X3
from .ops import Pair

_ZERO: Dict[torch.device, torch.Tensor] = {}


def zero_page(dev) -> torch.Tensor:
    z = _ZERO.get(dev)
    if z is None:
        z = _ZERO[dev] = torch.zeros(512, dtype=torch.bfloat16, device=dev)
        pipeline.publish()
    return z


def geom(B, H, W, C, kh=3, kw=3, stride=1, pad=1, up=0) -> dict:
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    return dict(B=B, H=H, W=W, C=C, Ho=Ho, Wo=Wo, kh=kh, kw=kw, stride=stride, pad=pad, up=up)


def pack_weight(w: torch.Tensor, mode: int, precision: int, pad_cout_to: int = 0, pad_cin_to: int = 0) -> Pair:
    """[Cout,Cin,kh,kw] fp32 -> GEMM operand (mode 0: [Cout, T*Cin]; mode 1: [Cin, T*Cout], taps flipped).
    Channel padding (zeros) brings tiny heads (Cout = 1/3/4) to the kernels' channel multiples."""
    w = w.detach().float()
    Cout, Cin, kh, kw = w.shape
    if pad_cout_to and Cout < pad_cout_to:
        w = torch.cat([w, w.new_zeros(pad_cout_to - Cout, Cin, kh, kw)], 0)
        Cout = pad_cout_to
    if pad_cin_to and Cin < pad_cin_to:
        w = torch.cat([w, w.new_zeros(Cout, pad_cin_to - Cin, kh, kw)], 1)
        Cin = pad_cin_to
    w = w.contiguous()
    rows, K = (Cout, kh * kw * Cin) if mode == 0 else (Cin, kh * kw * Cout)
    out = ops.empty_pair((rows, K), precision, w.device)
    a = lib.ConvWeightPackArgs(lib.ptr(w), lib.ptr(out[0]), lib.ptr(out[1]), Cout, Cin, kh, kw, mode)
    lib.call("mvp_conv_weight_pack", a)
    return out


def conv_gemm(x: Pair, g: dict, wk: Pair, N: int, *, bias=None, act=lib.ACT_NONE, residual=None, residual2=None, out_f32=None,
              out: Optional[Pair] = None, out_mask=None, relu_mask=None, mask_mode=0, precision=PREC_BF16X3, ldo=None, lda=None, act_after_res=False,
              tile_policy: int = 0) -> None:
    """Implicit-GEMM convolution: Y[B*Ho*Wo, N] = act(im2col(x) · wkᵀ + bias) (+ residual (+ residual2)).  From two rounds of 256x256
    tiles on (the DPT probe's layers at 8x the token grid) the library runs it on the large-M ping-pong kernel (csrc/gemm_pp.hip,
    CONV mode; ``tile_policy`` = lib.TILES_NO_PP keeps the tile kernels: tests compare the two bit for bit)."""
    M = g["B"] * g["Ho"] * g["Wo"]
    K = g["kh"] * g["kw"] * g["C"]
    o_hi, o_lo = out if out is not None else (None, None)
    ldn = ldo if ldo is not None else N
    args = lib.GemmArgs(
        lib.ptr(x[0]), lib.ptr(x[1]), lib.ptr(wk[0]), lib.ptr(wk[1]), lib.ptr(bias), lib.ptr(residual), lib.ptr(out_f32),
        lib.ptr(o_hi), lib.ptr(o_lo), M, N, K, lda if lda is not None else g["C"], K, ldn, ldn, ldn, act, precision, 0, 0, 0, 0,
        1, g["H"], g["W"], g["C"], g["Ho"], g["Wo"], g["kh"], g["kw"], g["stride"], g["pad"], g["up"], lib.ptr(zero_page(x[0].device)),
        lib.ptr(relu_mask), lib.ptr(out_mask), ldn, mask_mode, lib.ptr(residual2), int(act_after_res))
    args.tile_policy = int(tile_policy)
    lib.call("mvp_gemm_bias_act_res", args)


def tn_splits(tiles: int, M: int) -> int:
    """Split-K factor of the TN weight-gradient kernel.  One 64-pixel stage is 64 KB of LDS, so 2 workgroups are resident per CU
    (512 slots): pick the number of rounds R in 1..4 and S = 512 R / tiles that minimise R x (k-tiles per block + fixed cost).
    Measured (tools/tn_bench.py, us, old rule -> this rule): 512->512 @28^2 223 -> 197, @64^2 511 -> 458, 256->256 @56^2 232 -> 185."""
    best, best_cost = 1, None
    for R in (1, 2, 3, 4):
        S = max(1, min(64, (512 * R) // max(tiles, 1), max(1, M // 256)))
        cost = R * ((M + S * 64 - 1) // (S * 64) + 6)
        if best_cost is None or cost < best_cost:
            best, best_cost = S, cost
    return best


def conv_dw(gp: Pair, ldg: int, x: Pair, ldx: int, g: dict, Cout: int, dw: torch.Tensor, *, accumulate=False, precision=PREC_BF16X3,
            splits: Optional[int] = None) -> None:
    """dw[Cout, C, kh, kw] (+)= sum_m G[m, :Cout]ᵀ · im2col(x)[m]   (TN GEMM over pixels, split-K)."""
    M = g["B"] * g["Ho"] * g["Wo"]
    T = g["kh"] * g["kw"]
    tiles = ((Cout + 127) // 128) * T * (g["C"] // 128)
    if splits is None:
        splits = tn_splits(tiles, M)
    ws = torch.empty(int(lib.load().mvp_gemm_tn_workspace_bytes(Cout, g["C"], g["kh"], g["kw"], splits)) // 4, dtype=torch.float32, device=dw.device)
    a = lib.GemmTnArgs(lib.ptr(gp[0]), lib.ptr(gp[1]), lib.ptr(x[0]), lib.ptr(x[1]), lib.ptr(ws), lib.ptr(dw), lib.ptr(zero_page(dw.device)),
                       M, Cout, g["C"], ldg, ldx, g["H"], g["W"], g["Ho"], g["Wo"], g["kh"], g["kw"], g["stride"], g["pad"], g["up"],
                       splits, int(accumulate), precision)
    lib.call("mvp_gemm_tn_conv", a)


def upsample_nearest(src: torch.Tensor, B, H, W, C, f, *, want_f32=True, want_pair=True, precision=PREC_BF16X3, backward=False):
    """forward: [B,H,W,C] -> [B,H*f,W*f,C];  backward: src = fine gradient -> coarse [B,H,W,C] block sums."""
    Ho, Wo = (H, W) if backward else (H * f, W * f)
    dst = torch.empty(B * Ho * Wo, C, dtype=torch.float32, device=src.device) if want_f32 else None
    pair = ops.empty_pair((B * Ho * Wo, C), precision, src.device) if want_pair else (None, None)
    a = lib.UpsampleClArgs(lib.ptr(src), lib.ptr(dst), lib.ptr(pair[0]), lib.ptr(pair[1]), B, H, W, C, f, int(backward))
    lib.call("mvp_upsample_nearest_cl", a)
    return dst, pair


def upconv3_forward(x: Pair, w: torch.Tensor, bias, B, H, W, f, *, act=lib.ACT_NONE, out: Optional[Pair] = None, out_mask=None, out_f32=None,
                    precision=PREC_BF16X3) -> None:
    """y = act(conv3x3(nearest_upsample(x, f), w, padding=1) + bias) WITHOUT the convolution over the fine grid: the 9 per-tap products
    of every COARSE pixel in one GEMM (x [B*H*W, C] pair · [9*Cout, C]ᵀ), then a gather-sum per fine pixel (csrc/conv.hip
    upconv3_gather_kernel) that writes the pair / mask / fp32 outputs [B*H*f*W*f, Cout]."""
    Cout, C = w.shape[0], w.shape[1]
    wt = ops.split_bf16(w.detach().float().permute(2, 3, 0, 1).reshape(9 * Cout, C).contiguous(), precision)  # [(ky, kx, co), ci]
    t = torch.empty(B * H * W, 9 * Cout, dtype=torch.float32, device=w.device)
    ops.gemm(x, wt, B * H * W, 9 * Cout, C, out_f32=t, precision=precision)
    o_hi, o_lo = out if out is not None else (None, None)
    lib.call("mvp_upconv3_fwd_gather", lib.UpconvGatherArgs(lib.ptr(t), lib.ptr(bias), lib.ptr(out_f32), lib.ptr(o_hi), lib.ptr(o_lo), lib.ptr(out_mask),
                                                            B, H, W, Cout, f, act))


def upconv3_grad_boxsum(g: torch.Tensor, B, H, W, C, f, precision=PREC_BF16X3) -> Pair:
    """Box sums of the fine-grid gradient ``g`` [B, H*f, W*f, C] of "nearest x f, then 3x3 conv" per COARSE pixel and tap -> bf16 pair
    [B*H*W, 9*C] (column tap*C + c): the left operand of the coarse-grid weight- and input-gradient GEMMs (csrc/conv.hip)."""
    out = ops.empty_pair((B * H * W, 9 * C), precision, g.device)
    lib.call("mvp_upconv3_grad_boxsum", lib.UpconvBoxsumArgs(lib.ptr(g), lib.ptr(out[0]), lib.ptr(out[1]), B, H, W, C, f))
    return out


def mask_split(src: torch.Tensor, mask, M, N, *, ldo=None, write_f32=False, precision=PREC_BF16X3, lds=None, ldm=None):
    """(src * mask) -> bf16 pair [M, ldo] (pad columns zero); optionally written back to src (fp32)."""
    ldo = ldo if ldo is not None else N
    pair = ops.empty_pair((M, ldo), precision, src.device)
    a = lib.MaskSplitArgs(lib.ptr(src), lib.ptr(mask), lib.ptr(src) if write_f32 else None, lib.ptr(pair[0]), lib.ptr(pair[1]), M, N,
                          lds if lds is not None else N, ldm if ldm is not None else N, ldo)
    lib.call("mvp_mask_split", a)
    return pair
