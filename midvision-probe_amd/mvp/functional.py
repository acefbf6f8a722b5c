"""Autograd-visible operators of the probe path.  Each Function's forward AND backward call
the HIP kernels through the C ABI; PyTorch only provides the tape, memory and streams.

  interpolate        F.interpolate replacement (nearest / bilinear / bicubic)   train_depth.py:114, train_snorm.py:110
  linear_head        probes.py:427-432 for kernel_size=1 (cat -> bilinear x4 -> conv1x1), evaluated as
                     conv1x1 at token resolution then bilinear x4 (the two commute; 16x less work)
  depth_bins / depth_sigmoid   probes.py:176-212
  depth_loss / angular_loss    evals/utils/losses.py:97-182
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import lib, ops, pipeline
from .lib import PREC_BF16, PREC_BF16X3
from .vit import PackedFeatures, lookup_pack

_MODES = {"nearest": lib.RESIZE_NEAREST, "bilinear": lib.RESIZE_BILINEAR, "bicubic": lib.RESIZE_BICUBIC}


def _need_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise lib.MvpError(f"{what}: the HIP path needs device tensors (there is no CPU fallback)")


# --------------------------------------------------------------------------- interpolate
class _Interpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo, mode, align, sh, sw):
        _need_cuda(x, "interpolate")
        x = x.contiguous().float()
        B, C, Hi, Wi = x.shape
        y = torch.empty(B, C, Ho, Wo, dtype=torch.float32, device=x.device)
        ops.resize(x, y, B * C, Hi, Wi, Ho, Wo, mode, align_corners=align, scale_h=sh, scale_w=sw)
        ctx.cfg = (B, C, Hi, Wi, Ho, Wo, mode, align, sh, sw)
        return y

    @staticmethod
    def backward(ctx, gy):
        B, C, Hi, Wi, Ho, Wo, mode, align, sh, sw = ctx.cfg
        gy = gy.contiguous().float()
        gx = torch.empty(B, C, Hi, Wi, dtype=torch.float32, device=gy.device)
        ops.resize(gy, gx, B * C, Hi, Wi, Ho, Wo, mode, align_corners=align, scale_h=sh, scale_w=sw, backward=True)
        return gx, None, None, None, None, None, None


def interpolate(x: torch.Tensor, size=None, scale_factor=None, mode: str = "nearest", align_corners: Optional[bool] = None) -> torch.Tensor:
    """Drop-in for torch.nn.functional.interpolate on NCHW fp32 (the modes the reference uses)."""
    if mode not in _MODES:
        raise NotImplementedError(f"interpolate mode {mode!r}")
    Hi, Wi = x.shape[-2:]
    sh = sw = 0.0
    if size is not None:
        Ho, Wo = (size, size) if isinstance(size, int) else tuple(int(s) for s in size)
    else:
        sf = (scale_factor, scale_factor) if not isinstance(scale_factor, (tuple, list)) else scale_factor
        sh, sw = float(sf[0]), float(sf[1])
        Ho, Wo = int(Hi * sh), int(Wi * sw)  # floor, as torch
    return _Interpolate.apply(x, Ho, Wo, _MODES[mode], bool(align_corners), sh, sw)


def resize_antialias(x: torch.Tensor, size) -> torch.Tensor:
    """torchvision ``transforms.Resize(size)`` on a float tensor = ``F.interpolate(x, size, mode="bilinear",
    align_corners=False, antialias=True)`` (dino_res50.py:80,85).  Forward only (frozen-backbone input side)."""
    _need_cuda(x, "resize_antialias")
    if x.requires_grad:
        raise lib.MvpError("resize_antialias is forward-only (input side of the frozen backbone)")
    x = x.contiguous().float()
    B, C, Hi, Wi = x.shape
    Ho, Wo = (size, size) if isinstance(size, int) else tuple(int(v) for v in size)
    y = torch.empty(B, C, Ho, Wo, dtype=torch.float32, device=x.device)
    a = lib.ResizeArgs(lib.ptr(x), lib.ptr(y), B * C, Hi, Wi, Ho, Wo, lib.RESIZE_BILINEAR, 0, 0, 0, 0.0, 0.0)
    lib.call("mvp_resize_aa_fwd", a)
    return y


# --------------------------------------------------------------------------- linear head (k = 1)
def pack_features(feats: Sequence[torch.Tensor], precision: int) -> PackedFeatures:
    """Token-major bf16 operands for the head GEMMs: reuse the packing the backbone wrote next
    to these very maps when available, else pack the NCHW tensors now."""
    pack = lookup_pack(feats)
    if pack is not None and pack.precision == precision:
        return pack
    B, _, h, w = feats[0].shape
    for f in feats:
        _need_cuda(f, "probe features")
        if f.shape[0] != B or f.shape[-2:] != (h, w):
            raise lib.MvpError("linear head: all feature maps must share batch and resolution (reference quirk Q7)")
    Ctot = sum(int(f.shape[1]) for f in feats)
    pack = PackedFeatures(B, h, w, Ctot, precision, feats[0].device)
    off = 0
    for f in feats:
        C = int(f.shape[1])
        ops.pack_nchw_tokens(f.contiguous().float(), B, C, h * w, tok=pack.tok, ld_tok=pack.Cpad, col_off=off)
        off += C
    return pack


def _grad_dst(param, shape):
    """The parameter's slot in FlatAdamW's flat gradient buffer as a fresh [shape] view, or None (plain torch optimisers)."""
    from .optim import grad_destination

    d = grad_destination(param) if param is not None else None
    return d.view(shape) if d is not None and d.numel() == int(torch.Size(shape).numel()) else None


def _head_weight_grad(gl0: torch.Tensor, K: int, pack: PackedFeatures, pr: int, dst: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW[K, Ctot] = gl0ᵀ · F over the token axis: TN split-K kernel on the row-major gradient and the packed features
    (the weight gradient of the 1x1 conv, probes.py:431)."""
    from . import conv
    dev = gl0.device
    Kg = (K + 127) // 128 * 128
    if Kg != K:  # the TN tile is 128 output channels wide: zero pad columns
        gpad = gl0.new_zeros(pack.M, Kg)
        gpad[:, :K] = gl0
        gl0 = gpad
    gp = ops.split_bf16(gl0, pr)
    dW = dst if (dst is not None and pack.Cpad == pack.Ctot and tuple(dst.shape) == (K, pack.Cpad)) else torch.empty(K, pack.Cpad, dtype=torch.float32, device=dev)
    geo = dict(B=pack.B, H=pack.h, W=pack.w, C=pack.Cpad, Ho=pack.h, Wo=pack.w, kh=1, kw=1, stride=1, pad=0, up=0)
    conv.conv_dw(gp, Kg, pack.tok, pack.Cpad, geo, K, dW, precision=pr)
    return dW[:, :pack.Ctot]


class _LinearHeadK1(torch.autograd.Function):
    """logits_q[B, 4h, 4w, K] (channels-last) = bilinear_x4( F · Wᵀ + b )."""

    @staticmethod
    def forward(ctx, weight, bias, pack: PackedFeatures, precision: int):
        K, Ctot = weight.shape[0], pack.Ctot
        K4 = (K + 3) // 4 * 4
        dev = weight.device
        w2 = weight.detach().reshape(K, Ctot).float()
        b1 = bias.detach().float()
        if K4 != K or pack.Cpad != Ctot:
            # pad output channels to a multiple of 4 (channels-last kernels) and K to a multiple of 64 (GEMM)
            wpad = w2.new_zeros(K4, pack.Cpad)
            wpad[:K, :Ctot] = w2
            w2 = wpad
            b1 = torch.cat([b1, b1.new_zeros(K4 - K)], 0)
        wp = ops.split_bf16(w2.contiguous(), precision)
        l0 = torch.empty(pack.M, K4, dtype=torch.float32, device=dev)
        ops.gemm(pack.tok, wp, pack.M, K4, pack.Cpad, bias=b1.contiguous(), out_f32=l0, precision=precision)
        B, h, w = pack.B, pack.h, pack.w
        lq = torch.empty(B, 4 * h, 4 * w, K4, dtype=torch.float32, device=dev)
        ops.resize(l0, lq, B, h, w, 4 * h, 4 * w, lib.RESIZE_BILINEAR, channels_last=True, Cdim=K4, scale_h=4.0, scale_w=4.0)
        ctx.pack, ctx.precision, ctx.K, ctx.K4 = pack, precision, K, K4
        ctx.wshape = weight.shape
        ctx.generation = pack.generation
        return lq

    @staticmethod
    def backward(ctx, glq):
        pack, pr, K, K4 = ctx.pack, ctx.precision, ctx.K, ctx.K4
        B, h, w, Ctot = pack.B, pack.h, pack.w, pack.Ctot
        dev = glq.device
        if pack.generation != ctx.generation:
            raise lib.MvpError("linear head backward: the backbone ran again before this backward and overwrote the packed "
                               "features of this step (call backward before the next model(images))")
        glq = glq.contiguous()
        gl0 = torch.empty(pack.M, K4, dtype=torch.float32, device=dev)
        ops.resize(glq, gl0, B, h, w, 4 * h, 4 * w, lib.RESIZE_BILINEAR, channels_last=True, Cdim=K4, scale_h=4.0, scale_w=4.0, backward=True)
        dW = _head_weight_grad(gl0, K4, pack, pr)
        db = torch.empty(K4, dtype=torch.float32, device=dev)
        ops.colsum(gl0, db, pack.M, K4)
        return dW[:K].reshape(ctx.wshape), db[:K], None, None


def linear_head_k1(feats: Sequence[torch.Tensor], weight: torch.Tensor, bias: torch.Tensor, precision: int) -> torch.Tensor:
    """Returns channels-last logits [B, 4h, 4w, K4] (K4 = K rounded up to 4; extra channels are zero)."""
    pack = pack_features(list(feats), precision)
    return _LinearHeadK1.apply(weight, bias, pack, precision)


def linear_bins_forward(weight, bias, pack: PackedFeatures, precision: int, min_depth: float, max_depth: float):
    """The launches of ``_LinearBinsHead.forward`` (no tape): -> (depth [B,1,4h,4w], inv, gate).  Shared by the autograd Function and
    by the tape-free probe step (mvp/fused_step.py), so both issue the same kernels with the same arguments."""
    K, Ctot = weight.shape[0], pack.Ctot
    dev = weight.device
    w2 = weight.detach().reshape(K, Ctot).float()
    if pack.Cpad != Ctot:
        wpad = w2.new_zeros(K, pack.Cpad)
        wpad[:, :Ctot] = w2
        w2 = wpad
    wp = ops.split_bf16(w2.contiguous(), precision)
    l0 = torch.empty(pack.M, K, dtype=torch.float32, device=dev)
    ops.gemm(pack.tok, wp, pack.M, K, pack.Cpad, bias=bias.detach().float().contiguous(), out_f32=l0, precision=precision)
    B, h, w = pack.B, pack.h, pack.w
    P = B * 16 * h * w
    depth = torch.empty(B, 1, 4 * h, 4 * w, dtype=torch.float32, device=dev)
    inv = torch.empty(P, dtype=torch.float32, device=dev)
    gate = torch.empty(P, K // 8, dtype=torch.uint8, device=dev)
    a = lib.LinearBinsArgs(lib.ptr(l0), lib.ptr(depth), lib.ptr(inv), lib.ptr(gate), None, None, B, h, w, K, 4, min_depth, max_depth)
    lib.call("mvp_linear_bins_fwd", a)
    return depth, inv, gate


def linear_bins_backward(gd, depth, inv, gate, pack: PackedFeatures, precision: int, K: int, min_depth: float, max_depth: float,
                         dw_dst: Optional[torch.Tensor], db_dst: Optional[torch.Tensor]):
    """The launches of ``_LinearBinsHead.backward``: -> (dW [K, Ctot], db [K]); written into ``dw_dst`` / ``db_dst`` (the parameters'
    slots of FlatAdamW's flat gradient) when given and the packing has no channel padding."""
    B, h, w = pack.B, pack.h, pack.w
    dev = gd.device
    gl0 = torch.empty(pack.M, K, dtype=torch.float32, device=dev)
    a = lib.LinearBinsArgs(None, lib.ptr(depth), lib.ptr(inv), lib.ptr(gate), lib.ptr(gd.contiguous().float()), lib.ptr(gl0), B, h, w, K, 4,
                           min_depth, max_depth)
    lib.call("mvp_linear_bins_bwd", a)
    dW = _head_weight_grad(gl0, K, pack, precision, dst=dw_dst)
    db = db_dst
    if db is None:
        db = torch.empty(K, dtype=torch.float32, device=dev)
    ops.colsum(gl0, db, pack.M, K)
    return dW, db


class _LinearBinsHead(torch.autograd.Function):
    """DepthHead(linear, k=1, bindepth) in one piece (probes.py:153-157,176-200,427-432):
    GEMM at token resolution -> fused [bilinear x4 + bin expectation] kernel that keeps one gate
    bit per logit instead of the 16x larger upsampled logits -> depth [B,1,4h,4w]."""

    @staticmethod
    def forward(ctx, weight, bias, pack: PackedFeatures, precision: int, n_bins: int, min_depth: float, max_depth: float):
        depth, inv, gate = linear_bins_forward(weight, bias, pack, precision, min_depth, max_depth)
        ctx.pack, ctx.precision, ctx.cfg = pack, precision, (weight.shape[0], min_depth, max_depth)
        ctx.wshape, ctx.generation = weight.shape, pack.generation
        ctx.params = (weight, bias)
        ctx.save_for_backward(depth, inv, gate)
        return depth

    @staticmethod
    def backward(ctx, gd):
        depth, inv, gate = ctx.saved_tensors
        pack, pr = ctx.pack, ctx.precision
        K, mn, mx = ctx.cfg
        if pack.generation != ctx.generation:
            raise lib.MvpError("linear head backward: the backbone ran again before this backward and overwrote the packed "
                               "features of this step (call backward before the next model(images))")
        wparam, bparam = ctx.params
        dW, db = linear_bins_backward(gd, depth, inv, gate, pack, pr, K, mn, mx, _grad_dst(wparam, (K, pack.Ctot)), _grad_dst(bparam, (K,)))
        return dW.reshape(ctx.wshape), db, None, None, None, None, None


def linear_bins_head(feats: Sequence[torch.Tensor], weight, bias, precision: int, n_bins: int, min_depth: float, max_depth: float) -> torch.Tensor:
    pack = pack_features(list(feats), precision)
    return _LinearBinsHead.apply(weight, bias, pack, precision, n_bins, float(min_depth), float(max_depth))


class _LinearHeadKxK(torch.autograd.Function):
    """probes.py:427-432 for kernel_size > 1: the conv no longer commutes with the resample, so the
    features ARE upsampled (bilinear x4, as the reference does) into a channels-last bf16 pair and the
    k x k conv runs as an implicit GEMM; weight gradient through the TN split-K kernel."""

    @staticmethod
    def forward(ctx, weight, bias, feats, precision):
        from . import conv as cv

        B, _, h, w = feats[0].shape
        dev = weight.device
        K, Ctot, k, _ = weight.shape
        if Ctot % 128:
            raise lib.MvpError("Linear(k>1) on the HIP path needs sum(feat_dim) % 128 == 0")
        H4, W4 = 4 * h, 4 * w
        M = B * H4 * W4
        up = ops.empty_pair((M, Ctot), precision, dev)
        off = 0
        for f in feats:  # bilinear x4 per map (planar kernel), packed channels-last at its channel offset
            C = int(f.shape[1])
            big = torch.empty(B, C, H4, W4, dtype=torch.float32, device=dev)
            ops.resize(f.contiguous().float(), big, B * C, h, w, H4, W4, lib.RESIZE_BILINEAR, scale_h=4.0, scale_w=4.0)
            ops.pack_nchw_tokens(big, B, C, H4 * W4, tok=up, ld_tok=Ctot, col_off=off)
            off += C
        K4 = (K + 3) // 4 * 4
        g = cv.geom(B, H4, W4, Ctot, k, k, 1, k // 2)
        b4 = torch.cat([bias.detach().float(), bias.new_zeros(K4 - K).float()]) if K4 != K else bias.detach().float().contiguous()
        lq = torch.empty(B, H4, W4, K4, dtype=torch.float32, device=dev)
        cv.conv_gemm(up, g, cv.pack_weight(weight, 0, precision, pad_cout_to=K4), K4, bias=b4, out_f32=lq, precision=precision)
        ctx.up, ctx.g, ctx.cfg = up, g, (K, K4, M, precision)
        ctx.wshape = weight.shape
        return lq

    @staticmethod
    def backward(ctx, glq):
        from . import conv as cv

        K, K4, M, pr = ctx.cfg
        dev = glq.device
        gl = glq.contiguous().float().reshape(M, K4)
        LG = (K4 + 127) // 128 * 128
        gP = cv.mask_split(gl, None, M, K4, ldo=LG, precision=pr)
        dW = torch.empty(ctx.wshape, dtype=torch.float32, device=dev)
        cv.conv_dw(gP, LG, ctx.up, ctx.g["C"], ctx.g, K, dW, precision=pr)
        db = torch.empty(K4, dtype=torch.float32, device=dev)
        ops.colsum(gl, db, M, K4)
        return dW, db[:K].contiguous(), None, None


def linear_head_kxk(feats: Sequence[torch.Tensor], weight: torch.Tensor, bias: torch.Tensor, precision: int) -> torch.Tensor:
    for f in feats:
        _need_cuda(f, "probe features")
    return _LinearHeadKxK.apply(weight, bias, list(feats), precision)


class _ConvValid(torch.autograd.Function):
    """An UN-PADDED k x k convolution of an NCHW fp32 map (nn.Conv2d(cin, cout, k), what probes.py:400-412 ``make_conv`` builds), on the
    implicit-GEMM kernels: the map is packed channels-last into a bf16 pair, Y = im2col(x) · Wᵀ + b (csrc/gemm.hip conv mode, pad 0);
    backward: dW by the TN split-K kernel over pixels, dX = the "full" correlation of the output gradient with the flipped,
    transposed taps (the same conv kernel at pad k-1), db a column sum.  Used by MultiscaleHead(kernel_size > 1)."""

    @staticmethod
    def forward(ctx, x, weight, bias, precision, need_dx):
        from . import conv as cv

        _need_cuda(x, "conv2d_valid")
        B, C, H, W = x.shape
        Cout, Cin, k, k2 = weight.shape
        if Cin != C or k != k2 or H < k or W < k:
            raise lib.MvpError(f"conv2d_valid: input {tuple(x.shape)} does not fit a {tuple(weight.shape)} kernel")
        if C % 128:
            raise lib.MvpError(f"conv2d_valid on the HIP path needs input channels % 128 == 0 (got {C})")
        dev = x.device
        tok = ops.empty_pair((B * H * W, C), precision, dev)
        ops.pack_nchw_tokens(x.contiguous().float(), B, C, H * W, tok=tok, ld_tok=C, col_off=0)
        g = cv.geom(B, H, W, C, k, k, 1, 0)
        N4 = (Cout + 3) // 4 * 4
        b4 = bias.detach().float().contiguous()
        if N4 != Cout:
            b4 = torch.cat([b4, b4.new_zeros(N4 - Cout)])
        y = torch.empty(B, g["Ho"], g["Wo"], N4, dtype=torch.float32, device=dev)
        cv.conv_gemm(tok, g, cv.pack_weight(weight, 0, precision, pad_cout_to=N4), N4, bias=b4, out_f32=y, precision=precision)
        ctx.tok, ctx.g, ctx.cfg = tok, g, (Cout, N4, precision, bool(need_dx))
        ctx.save_for_backward(weight)
        return y[..., :Cout].permute(0, 3, 1, 2)  # NCHW view of the channels-last result

    @staticmethod
    def backward(ctx, gy):
        from . import conv as cv

        (weight,) = ctx.saved_tensors
        Cout, N4, pr, need_dx = ctx.cfg
        g = ctx.g
        B, H, W, C, Ho, Wo, k = g["B"], g["H"], g["W"], g["C"], g["Ho"], g["Wo"], g["kh"]
        dev = gy.device
        M = B * Ho * Wo
        gl = torch.zeros(M, N4, dtype=torch.float32, device=dev)
        gl[:, :Cout] = gy.permute(0, 2, 3, 1).reshape(M, Cout)
        LG = (N4 + 127) // 128 * 128
        gP = cv.mask_split(gl, None, M, N4, ldo=LG, precision=pr)
        dW = torch.empty(weight.shape, dtype=torch.float32, device=dev)
        cv.conv_dw(gP, LG, ctx.tok, C, g, Cout, dW, precision=pr)
        db = torch.empty(N4, dtype=torch.float32, device=dev)
        ops.colsum(gl, db, M, N4)
        dx = None
        if need_dx:
            wT = cv.pack_weight(weight, 1, pr, pad_cout_to=LG)  # [Cin, taps (flipped) x Cout]
            dxc = torch.empty(B, H, W, C, dtype=torch.float32, device=dev)
            cv.conv_gemm(gP, cv.geom(B, Ho, Wo, LG, k, k, 1, k - 1), wT, C, out_f32=dxc, precision=pr)
            dx = dxc.permute(0, 3, 1, 2)
        return dx, dW, db[:Cout].contiguous(), None, None


def conv2d_valid(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, precision: int, need_input_grad: bool = True) -> torch.Tensor:
    """``F.conv2d(x, weight, bias)`` (no padding, stride 1) for NCHW fp32 device maps, differentiable in weight, bias and (unless
    ``need_input_grad`` is False: frozen features) the input."""
    return _ConvValid.apply(x, weight, bias, precision, need_input_grad)


# --------------------------------------------------------------------------- depth predictors
class _DepthPredict(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits_cl, K, min_depth, max_depth, kind):
        B, H, W, Kp = logits_cl.shape
        if Kp != K and not (kind == 1 and K == 1):
            raise lib.MvpError("depth predictor: channel padding is only supported for sigmoid heads")
        P = B * H * W
        lg = logits_cl.contiguous()
        if kind == 1 and Kp != 1:
            lg = lg[..., :1].contiguous()
        depth = torch.empty(B, 1, H, W, dtype=torch.float32, device=lg.device)
        inv = torch.empty(P, dtype=torch.float32, device=lg.device) if kind == 0 else None
        ops.depth_predict_fwd(lg, depth, inv, P, K, min_depth, max_depth, kind)
        ctx.save_for_backward(lg, depth, inv if inv is not None else depth)
        ctx.cfg = (P, K, Kp, min_depth, max_depth, kind, logits_cl.shape)
        return depth

    @staticmethod
    def backward(ctx, gd):
        lg, depth, inv = ctx.saved_tensors
        P, K, Kp, mn, mx, kind, shape = ctx.cfg
        gl = torch.empty_like(lg)
        ops.depth_predict_bwd(lg, depth, inv if kind == 0 else None, gd.contiguous().float(), gl, P, K, mn, mx, kind)
        if kind == 1 and Kp != 1:
            full = gl.new_zeros(shape)
            full[..., :1] = gl
            gl = full
        return gl, None, None, None, None


def depth_bins(logits_cl, n_bins, min_depth, max_depth):
    return _DepthPredict.apply(logits_cl, n_bins, float(min_depth), float(max_depth), 0)


def depth_sigmoid(logits_cl, min_depth, max_depth):
    return _DepthPredict.apply(logits_cl, 1, float(min_depth), float(max_depth), 1)


# --------------------------------------------------------------------------- losses
_ONE = {}


def _one(dev) -> torch.Tensor:
    t = _ONE.get(dev)
    if t is None:
        t = _ONE[dev] = torch.ones((), dtype=torch.float32, device=dev)
        pipeline.publish()
    return t


def backward(loss: torch.Tensor) -> None:
    """``loss.backward()`` with a cached device scalar 1 as the root gradient: no ones_like fill, and the loss Functions
    recognise it and hand their stored gradient on without a multiply kernel."""
    torch.autograd.backward(loss, grad_tensors=_one(loss.device))


def _scaled(grad: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    one = _ONE.get(g.device)
    if one is not None and g.data_ptr() == one.data_ptr():
        return grad
    return grad * g


class _DepthLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, w_sig, w_grad, max_depth):
        _need_cuda(pred, "DepthLoss")
        if not (target.is_cuda and target.is_contiguous() and target.dtype == torch.float32):
            raise lib.MvpError("DepthLoss: target must be a contiguous fp32 device tensor (it is modified in place, quirk Q2)")
        B = pred.shape[0]
        HW = pred.numel() // B
        p = pred.contiguous().float()
        out = torch.empty(4, dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p)
        ws = torch.empty(ops.depth_loss_workspace_bytes(B, HW) // 4 + 4, dtype=torch.float32, device=p.device)
        ops.depth_loss(p, target, out, grad, ws, B, HW, w_sig, w_grad, max_depth)
        ctx.save_for_backward(grad)
        ctx.parts = out
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return _scaled(grad, g), None, None, None, None


def depth_loss(pred, target, w_sig=10.0, w_grad=0.5, max_depth=10.0):
    return _DepthLoss.apply(pred, target, float(w_sig), float(w_grad), float(max_depth))


class _AngularLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, mask, eps):
        _need_cuda(pred, "angular_loss")
        B, Cp = pred.shape[:2]
        HW = pred.numel() // (B * Cp)
        p = pred.contiguous().float()
        m = mask.reshape(B, -1).to(torch.uint8).contiguous()
        out = torch.empty(4, dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p)
        ws = torch.empty(1024, dtype=torch.float32, device=p.device)
        ops.angular_loss(p, gt.contiguous().float(), m, out, grad, ws, B, Cp, HW, eps)
        ctx.save_for_backward(grad)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return _scaled(grad, g), None, None, None


def angular_loss(pred, gt, mask, uncertainty_aware=False, eps=1e-4):
    assert mask.ndim == 4, f"mask should be (batch x height x width) not {mask.shape}"
    assert pred.shape[1] == (4 if uncertainty_aware else 3)
    return _AngularLoss.apply(pred, gt, mask, float(eps))
