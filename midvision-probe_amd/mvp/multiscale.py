"""MultiscaleHead (evals/models/probes.py:435-458, DepthHead's default head_type) on the HIP kernels, kernel_size = 1.

Reference graph:  f_i = conv_i(feat_i) (1x1)  ->  bilinear to the last map's size  ->  cat, ReLU  ->  bilinear x2  ->
conv_mid (1x1, ReLU, 1x1, ReLU, 1x1), ReLU  ->  bilinear x4  ->  conv_out (1x1, ReLU, 1x1).

A 1x1 convolution and a bilinear resample commute (both linear, resampling weights sum to 1, so the bias commutes too); ReLU
does not.  The three places where a conv directly follows a resample are therefore evaluated conv-first, at the LOWER
resolution (4x resp. 16x fewer GEMM rows), exactly like the k=1 linear probe:
    a = relu(cat_i conv_i(resize(feat_i)))        [M0, 4*Hd]      (feature maps are resampled before the conv)
    b = conv_mid.0(a)                             [M0, Hd]
    c = relu(up2(b))                              [M1 = 4*M0, Hd]
    d = relu(conv_mid.2(c));  e = relu(conv_mid.4(d))
    f = conv_out.0(e)                             [M1, Hd]
    g = relu(up4(f))                              [M2 = 16*M1, Hd]
    logits = conv_out.2(g)                        [M2, K]
Everything is channels-last; GEMMs are the bf16-pair MFMA kernel with fused bias / ReLU / gate-mask epilogues, weight gradients
the TN split-K kernel, resamples the channels-last resize kernel and its adjoint.  One autograd.Function for the head."""
from __future__ import annotations

from typing import List, Optional

import torch

from . import conv as cv
from . import lib, ops
from .lib import ACT_NONE, ACT_RELU
from .vit import PackedFeatures


def _up(n, m):
    return (n + m - 1) // m * m


def multiscale_param_list(head) -> List[torch.Tensor]:
    ps = []
    for c in head.convs:
        ps += [c.weight, c.bias]
    for i in (0, 2, 4):
        ps += [head.conv_mid[i].weight, head.conv_mid[i].bias]
    for i in (0, 2):
        ps += [head.conv_out[i].weight, head.conv_out[i].bias]
    return ps


def _relu_split(src, M, N, pr):
    """max(src, 0) -> bf16 pair + byte gate (src is overwritten with the rectified values)."""
    pair = ops.empty_pair((M, N), pr, src.device)
    mask = torch.empty(M, N, dtype=torch.uint8, device=src.device)
    a = lib.MaskSplitArgs(lib.ptr(src), None, None, lib.ptr(pair[0]), lib.ptr(pair[1]), M, N, N, N, N, lib.ptr(mask))
    lib.call("mvp_mask_split", a)
    return pair, mask


class _Multiscale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pack: PackedFeatures, dims, precision: int, *params):
        pr, dev = precision, params[0].device
        B, h, w = pack.B, pack.h, pack.w
        nf = len(dims)
        Hd = params[0].shape[0]
        Cout = params[-2].shape[0]
        if Hd % 128 or any(c % 128 for c in dims):
            raise lib.MvpError(f"MultiscaleHead on the HIP path needs feature and hidden channels that are multiples of 128 (got {dims}, {Hd})")
        K4 = _up(Cout, 4)
        M0, M1, M2 = B * h * w, B * 4 * h * w, B * 64 * h * w
        det = [p.detach().float() for p in params]
        W = lambda i: det[2 * i].reshape(det[2 * i].shape[0], -1).contiguous()  # noqa: E731
        bvec = lambda i: det[2 * i + 1].contiguous()  # noqa: E731
        CH = nf * Hd
        aP = ops.empty_pair((M0, CH), pr, dev)
        ma = torch.empty(M0, CH, dtype=torch.uint8, device=dev)
        off = 0
        for i, C in enumerate(dims):
            x = (pack.tok[0][:, off:], pack.tok[1][:, off:] if pack.tok[1] is not None else None)
            o = (aP[0][:, i * Hd:], aP[1][:, i * Hd:] if aP[1] is not None else None)
            ops.gemm(x, ops.split_bf16(W(i), pr), M0, Hd, C, bias=bvec(i), act=ACT_RELU, out=o, ldob=CH, out_mask=ma[:, i * Hd:], ldm=CH,
                     precision=pr, lda=pack.Cpad)
            off += C
        j = nf
        b = torch.empty(M0, Hd, dtype=torch.float32, device=dev)
        ops.gemm(aP, ops.split_bf16(W(j), pr), M0, Hd, CH, bias=bvec(j), out_f32=b, precision=pr, splitk=1)
        c32 = torch.empty(M1, Hd, dtype=torch.float32, device=dev)
        ops.resize(b, c32, B, h, w, 2 * h, 2 * w, lib.RESIZE_BILINEAR, channels_last=True, Cdim=Hd, scale_h=2.0, scale_w=2.0)
        cP, mc = _relu_split(c32, M1, Hd, pr)
        dP, md = ops.empty_pair((M1, Hd), pr, dev), torch.empty(M1, Hd, dtype=torch.uint8, device=dev)
        ops.gemm(cP, ops.split_bf16(W(j + 1), pr), M1, Hd, Hd, bias=bvec(j + 1), act=ACT_RELU, out=dP, out_mask=md, precision=pr)
        eP, me = ops.empty_pair((M1, Hd), pr, dev), torch.empty(M1, Hd, dtype=torch.uint8, device=dev)
        ops.gemm(dP, ops.split_bf16(W(j + 2), pr), M1, Hd, Hd, bias=bvec(j + 2), act=ACT_RELU, out=eP, out_mask=me, precision=pr)
        f = c32  # reuse
        ops.gemm(eP, ops.split_bf16(W(j + 3), pr), M1, Hd, Hd, bias=bvec(j + 3), out_f32=f, precision=pr, splitk=1)
        g32 = torch.empty(M2, Hd, dtype=torch.float32, device=dev)
        ops.resize(f, g32, B, 2 * h, 2 * w, 8 * h, 8 * w, lib.RESIZE_BILINEAR, channels_last=True, Cdim=Hd, scale_h=4.0, scale_w=4.0)
        gP, mg = _relu_split(g32, M2, Hd, pr)
        del g32
        w_last, b_last = W(j + 4), bvec(j + 4)
        if K4 != Cout:
            w_last = torch.cat([w_last, w_last.new_zeros(K4 - Cout, Hd)], 0)
            b_last = torch.cat([b_last, b_last.new_zeros(K4 - Cout)], 0)
        logits = torch.empty(B, 8 * h, 8 * w, K4, dtype=torch.float32, device=dev)
        ops.gemm(gP, ops.split_bf16(w_last.contiguous(), pr), M2, K4, Hd, bias=b_last.contiguous(), out_f32=logits, precision=pr, splitk=1)
        ctx.pack, ctx.pr, ctx.cfg = pack, pr, (B, h, w, tuple(dims), Hd, Cout, K4)
        ctx.acts = (aP, ma, cP, mc, dP, md, eP, me, gP, mg)
        ctx.generation = pack.generation
        ctx.save_for_backward(*params)
        return logits

    @staticmethod
    def backward(ctx, g_logits):
        pack, pr = ctx.pack, ctx.pr
        B, h, w, dims, Hd, Cout, K4 = ctx.cfg
        aP, ma, cP, mc, dP, md, eP, me, gP, mg = ctx.acts
        params = ctx.saved_tensors
        det = [p.detach().float() for p in params]
        dev = g_logits.device
        if pack.generation != ctx.generation:
            raise lib.MvpError("MultiscaleHead backward: the backbone ran again before this backward and overwrote this step's packed features")
        nf, CH = len(dims), len(dims) * Hd
        M0, M1, M2 = B * h * w, B * 4 * h * w, B * 64 * h * w
        grads: List[Optional[torch.Tensor]] = [None] * len(params)
        j = nf

        def wgrad(idx, gPair, ldg, xPair, ldx, Cin, n_out, M, Hs, Ws):
            dW = torch.empty(params[2 * idx].shape, dtype=torch.float32, device=dev)
            cv.conv_dw(gPair, ldg, xPair, ldx, cv.geom(B, Hs, Ws, Cin, 1, 1, 1, 0), n_out, dW, precision=pr)
            grads[2 * idx] = dW

        def bgrad(idx, gF, M, N, n_true=None, ld=None):
            db = torch.empty(N, dtype=torch.float32, device=dev)
            ops.colsum(gF, db, M, N, ld=ld)
            grads[2 * idx + 1] = db if n_true is None else db[:n_true].contiguous()

        def dgrad(gPair, Kin, idx, n_in, mask, M, Hs, Ws, pad_to=0):
            """gradient wrt the conv input (gated by the ReLU that produced it): [M, n_in] fp32 + pair"""
            oF = torch.empty(M, n_in, dtype=torch.float32, device=dev)
            oP = ops.empty_pair((M, n_in), pr, dev)
            wT = cv.pack_weight(det[2 * idx].reshape(det[2 * idx].shape[0], -1, 1, 1), 1, pr, pad_cout_to=pad_to)
            cv.conv_gemm(gPair, cv.geom(B, Hs, Ws, Kin, 1, 1, 1, 0), wT, n_in, relu_mask=mask, mask_mode=2, out_f32=oF, out=oP, precision=pr)
            return oF, oP

        # conv_out.2
        gl = g_logits.contiguous().float().reshape(M2, K4)
        LG = _up(K4, 128)
        glP = cv.mask_split(gl, None, M2, K4, ldo=LG, precision=pr)
        wgrad(j + 4, glP, LG, gP, Hd, Hd, Cout, M2, 8 * h, 8 * w)
        bgrad(j + 4, gl, M2, K4, Cout)
        ggF, _ = dgrad(glP, LG, j + 4, Hd, mg, M2, 8 * h, 8 * w, pad_to=LG)
        # up4 adjoint, conv_out.0
        gf = torch.empty(M1, Hd, dtype=torch.float32, device=dev)
        ops.resize(ggF, gf, B, 2 * h, 2 * w, 8 * h, 8 * w, lib.RESIZE_BILINEAR, channels_last=True, Cdim=Hd, scale_h=4.0, scale_w=4.0, backward=True)
        del ggF
        gfP = ops.split_bf16(gf, pr)
        wgrad(j + 3, gfP, Hd, eP, Hd, Hd, Hd, M1, 2 * h, 2 * w)
        bgrad(j + 3, gf, M1, Hd)
        geF, geP = dgrad(gfP, Hd, j + 3, Hd, me, M1, 2 * h, 2 * w)
        # conv_mid.4, conv_mid.2
        wgrad(j + 2, geP, Hd, dP, Hd, Hd, Hd, M1, 2 * h, 2 * w)
        bgrad(j + 2, geF, M1, Hd)
        gdF, gdP = dgrad(geP, Hd, j + 2, Hd, md, M1, 2 * h, 2 * w)
        wgrad(j + 1, gdP, Hd, cP, Hd, Hd, Hd, M1, 2 * h, 2 * w)
        bgrad(j + 1, gdF, M1, Hd)
        gcF, _ = dgrad(gdP, Hd, j + 1, Hd, mc, M1, 2 * h, 2 * w)
        # up2 adjoint, conv_mid.0
        gb = torch.empty(M0, Hd, dtype=torch.float32, device=dev)
        ops.resize(gcF, gb, B, h, w, 2 * h, 2 * w, lib.RESIZE_BILINEAR, channels_last=True, Cdim=Hd, scale_h=2.0, scale_w=2.0, backward=True)
        gbP = ops.split_bf16(gb, pr)
        wgrad(j, gbP, Hd, aP, CH, CH, Hd, M0, h, w)
        bgrad(j, gb, M0, Hd)
        gaF, gaP = dgrad(gbP, Hd, j, CH, ma, M0, h, w)
        # the per-map 1x1 convs on the packed features
        off = 0
        for i, C in enumerate(dims):
            gs = (gaP[0][:, i * Hd:], gaP[1][:, i * Hd:] if gaP[1] is not None else None)
            x = (pack.tok[0][:, off:], pack.tok[1][:, off:] if pack.tok[1] is not None else None)
            wgrad(i, gs, CH, x, pack.Cpad, C, Hd, M0, h, w)
            bgrad(i, gaF[:, i * Hd:], M0, Hd, ld=CH)
            off += C
        return (None, None, None, *grads)


def multiscale_logits(pack: PackedFeatures, dims, head, precision: int) -> torch.Tensor:
    """Channels-last logits [B, 8h, 8w, K4] of MultiscaleHead (kernel_size 1)."""
    return _Multiscale.apply(pack, tuple(dims), precision, *multiscale_param_list(head))
