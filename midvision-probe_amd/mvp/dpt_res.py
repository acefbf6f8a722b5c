"""DPT probe head, CNN / ResNet-pyramid variant (evals/models/probes.py:215-399 with
``is_transformer=False``), on the HIP conv path.

Differences from the ViT variant (mvp/dpt.py): the four taps live at four resolutions
(s, 2s, 4s, 8s with s = coarsest), conv_i are 3x3 without bias, residual units are
PRE-activation with an in-place first ReLU (so the skip adds relu(x), probes.py:275,301-306),
and every fusion block ends in a bilinear x2 (align_corners=True) resample (probes.py:255-258).

Only relu(.) of each residual-unit input is ever consumed downstream, so the producing GEMM
applies that ReLU in its epilogue (after the skip adds where needed) and emits the byte gate.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import conv as cv
from . import lib, ops
from .lib import ACT_NONE, ACT_RELU

RCU_ORDER = [(3, 2), (2, 1), (2, 2), (1, 1), (1, 2), (0, 1), (0, 2)]


def dpt_res_param_list(head) -> List[torch.Tensor]:
    ps = [getattr(head, f"conv_{i}").weight for i in range(4)]
    for blk, unit in RCU_ORDER:
        rcu = getattr(getattr(head, f"ref_{blk}"), f"resConfUnit{unit}")
        ps += [rcu.conv1.weight, rcu.conv1.bias, rcu.conv2.weight, rcu.conv2.bias]
    ps += [head.out_conv[0].weight, head.out_conv[0].bias, head.out_conv[2].weight, head.out_conv[2].bias]
    return ps


def _up(n, m):
    return (n + m - 1) // m * m


def _bilinear2(src, B, H, W, C, backward=False):
    """x2 bilinear, align_corners=True, channels-last fp32 (forward or adjoint)."""
    if backward:
        dst = torch.empty(B * H * W, C, dtype=torch.float32, device=src.device)
    else:
        dst = torch.empty(B * 4 * H * W, C, dtype=torch.float32, device=src.device)
    ops.resize(src, dst, B, H, W, 2 * H, 2 * W, lib.RESIZE_BILINEAR, align_corners=True, channels_last=True, Cdim=C, scale_h=2.0, scale_w=2.0,
               backward=backward)
    return dst


class _DPTRes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, toks, dims, B, precision, *params):
        """toks[i]: channels-last bf16 pair [B*H_i*W_i, C_i] of tap i; dims[i] = (C_i, H_i, W_i)."""
        pr, dev = precision, params[0].device
        det = [p.detach() for p in params]
        Hd = det[0].shape[0]
        Cout = det[-2].shape[0]
        K4 = _up(Cout, 4)
        if Hd % 128 or any(d[0] % 128 for d in dims):
            raise lib.MvpError("DPT(ResNet) on the HIP path needs channel counts that are multiples of 128")
        for i in range(3):  # each level must be exactly 2x the next (skip shapes must match, probes.py:246)
            assert dims[i][1] == 2 * dims[i + 1][1] and dims[i][2] == 2 * dims[i + 1][2], "Shape of skip_x must match x"

        # r_i = relu(conv_i(feat_i)), 3x3 no bias
        r, mf = [], []
        for i in range(4):
            C, H, W = dims[i]
            M = B * H * W
            rF = torch.empty(M, Hd, dtype=torch.float32, device=dev)
            rP = ops.empty_pair((M, Hd), pr, dev)
            m = torch.empty(M, Hd, dtype=torch.uint8, device=dev)
            cv.conv_gemm(toks[i], cv.geom(B, H, W, C, 3, 3, 1, 1), cv.pack_weight(det[i], 0, pr), Hd, act=ACT_RELU, out_f32=rF, out=rP, out_mask=m, precision=pr)
            r.append((rF, rP))
            mf.append(m)

        saved = []

        def rcu(xF, xP, H, W, w1, b1, w2, b2, extra=None, relu_out=False):
            M = B * H * W
            g = cv.geom(B, H, W, Hd, 3, 3, 1, 1)
            aP = ops.empty_pair((M, Hd), pr, dev)
            ma = torch.empty(M, Hd, dtype=torch.uint8, device=dev)
            cv.conv_gemm(xP, g, cv.pack_weight(w1, 0, pr), Hd, bias=b1.float().contiguous(), act=ACT_RELU, out=aP, out_mask=ma, precision=pr)
            yF = torch.empty(M, Hd, dtype=torch.float32, device=dev)
            yP = ops.empty_pair((M, Hd), pr, dev)
            my = torch.empty(M, Hd, dtype=torch.uint8, device=dev) if relu_out else None
            cv.conv_gemm(aP, g, cv.pack_weight(w2, 0, pr), Hd, bias=b2.float().contiguous(), residual=xF, residual2=extra,
                         act=ACT_RELU if relu_out else ACT_NONE, act_after_res=relu_out, out_f32=yF, out=yP, out_mask=my, precision=pr)
            saved.append((xP, aP, ma, my, H, W))
            return yF, yP

        base = 4
        up = None  # bilinear x2 of the previous block's output (fp32)
        for n, (blk, unit) in enumerate(RCU_ORDER):
            w1, b1, w2, b2 = det[base + 4 * n: base + 4 * n + 4]
            _, H, W = dims[blk]
            if unit == 1:    # relu(RCU1(r_blk) + up): only the relu'd sum feeds RCU2
                cur = rcu(r[blk][0], r[blk][1], H, W, w1, b1, w2, b2, extra=up, relu_out=True)
            else:
                src = r[3] if blk == 3 else cur
                y = rcu(src[0], src[1], H, W, w1, b1, w2, b2)
                up = _bilinear2(y[0], B, H, W, Hd)
        H2, W2 = 2 * dims[0][1], 2 * dims[0][2]
        M2 = B * H2 * W2
        upP = cv.mask_split(up, None, M2, Hd, precision=pr)

        w0, b0, w2o, b2o = det[-4:]
        g3 = cv.geom(B, H2, W2, Hd, 3, 3, 1, 1)
        h0P = ops.empty_pair((M2, Hd), pr, dev)
        m0 = torch.empty(M2, Hd, dtype=torch.uint8, device=dev)
        cv.conv_gemm(upP, g3, cv.pack_weight(w0, 0, pr), Hd, bias=b0.float().contiguous(), act=ACT_RELU, out=h0P, out_mask=m0, precision=pr)
        b2p = torch.cat([b2o.float(), b2o.new_zeros(K4 - Cout).float()]) if K4 != Cout else b2o.float().contiguous()
        logits = torch.empty(B, H2, W2, K4, dtype=torch.float32, device=dev)
        cv.conv_gemm(h0P, g3, cv.pack_weight(w2o, 0, pr, pad_cout_to=K4), K4, bias=b2p, out_f32=logits, precision=pr)

        ctx.toks, ctx.dims, ctx.B, ctx.pr, ctx.cfg = toks, dims, B, pr, (Hd, Cout, K4, H2, W2)
        ctx.saved, ctx.mf, ctx.upP, ctx.h0P, ctx.m0 = saved, mf, upP, h0P, m0
        ctx.save_for_backward(*params)
        return logits

    @staticmethod
    def backward(ctx, g_logits):
        pr, B, dims = ctx.pr, ctx.B, ctx.dims
        Hd, Cout, K4, H2, W2 = ctx.cfg
        params = ctx.saved_tensors
        det = [p.detach() for p in params]
        dev = g_logits.device
        M2 = B * H2 * W2
        grads: List[Optional[torch.Tensor]] = [None] * len(params)

        def new_like(p):
            return torch.empty(p.shape, dtype=torch.float32, device=dev)

        def bias_grad(gF, N, n_true=None):
            db = torch.empty(N, dtype=torch.float32, device=dev)
            ops.colsum(gF, db, gF.shape[0], N)
            return db if n_true is None else db[:n_true].contiguous()

        def gate_split(gF, mask, M, want_f32=True):
            """(gF * mask) -> fp32 (new tensor) + pair."""
            outF = torch.empty(M, Hd, dtype=torch.float32, device=dev) if want_f32 else None
            outP = ops.empty_pair((M, Hd), pr, dev)
            lib.call("mvp_mask_split", lib.MaskSplitArgs(lib.ptr(gF), lib.ptr(mask), lib.ptr(outF), lib.ptr(outP[0]), lib.ptr(outP[1]), M, Hd, Hd, Hd, Hd))
            return outF, outP

        # ---- out_conv
        w0, b0, w2o, b2o = det[-4:]
        g_logits = g_logits.contiguous().float().reshape(M2, K4)
        LG = _up(K4, 128)
        gP = cv.mask_split(g_logits, None, M2, K4, ldo=LG, precision=pr)
        g3 = cv.geom(B, H2, W2, Hd, 3, 3, 1, 1)
        grads[-2] = new_like(w2o)
        cv.conv_dw(gP, LG, ctx.h0P, Hd, g3, Cout, grads[-2], precision=pr)
        grads[-1] = bias_grad(g_logits, K4, Cout)
        gh0F = torch.empty(M2, Hd, dtype=torch.float32, device=dev)
        gh0P = ops.empty_pair((M2, Hd), pr, dev)
        cv.conv_gemm(gP, cv.geom(B, H2, W2, LG, 3, 3, 1, 1), cv.pack_weight(w2o, 1, pr, pad_cout_to=LG), Hd, relu_mask=ctx.m0, mask_mode=2,
                     out_f32=gh0F, out=gh0P, precision=pr)
        grads[-4] = new_like(w0)
        cv.conv_dw(gh0P, Hd, ctx.upP, Hd, g3, Hd, grads[-4], precision=pr)
        grads[-3] = bias_grad(gh0F, Hd)
        g_up = torch.empty(M2, Hd, dtype=torch.float32, device=dev)
        cv.conv_gemm(gh0P, g3, cv.pack_weight(w0, 1, pr), Hd, out_f32=g_up, precision=pr)
        del gh0F, gh0P

        def rcu_bwd(gy, gyP, sv, idx):
            """y = conv2(a) + xr (+extra), a = relu(conv1(xr)).  gy = dL/dy (fp32) and its pair.
            Returns dL/dxr = convT1(g_a) + gy."""
            xP, aP, ma, _, H, W = sv
            M = B * H * W
            g = cv.geom(B, H, W, Hd, 3, 3, 1, 1)
            w1, b1, w2, b2 = det[idx: idx + 4]
            grads[idx + 2] = new_like(w2)
            cv.conv_dw(gyP, Hd, aP, Hd, g, Hd, grads[idx + 2], precision=pr)
            grads[idx + 3] = bias_grad(gy, Hd)
            gaF = torch.empty(M, Hd, dtype=torch.float32, device=dev)
            gaP = ops.empty_pair((M, Hd), pr, dev)
            cv.conv_gemm(gyP, g, cv.pack_weight(w2, 1, pr), Hd, relu_mask=ma, mask_mode=2, out_f32=gaF, out=gaP, precision=pr)
            grads[idx] = new_like(w1)
            cv.conv_dw(gaP, Hd, xP, Hd, g, Hd, grads[idx], precision=pr)
            grads[idx + 1] = bias_grad(gaF, Hd)
            gx = torch.empty(M, Hd, dtype=torch.float32, device=dev)
            cv.conv_gemm(gaP, g, cv.pack_weight(w1, 1, pr), Hd, residual=gy, out_f32=gx, precision=pr)
            return gx

        base = 4
        g_r: List[Optional[torch.Tensor]] = [None] * 4   # dL/d r_i (before the relu gate of conv_i)
        for n in reversed(range(len(RCU_ORDER))):
            blk, unit = RCU_ORDER[n]
            sv = ctx.saved[n]
            _, H, W = dims[blk]
            M = B * H * W
            if unit == 2:
                # y = RCU2(x); its output was bilinearly upsampled: pull the gradient back first
                gy = _bilinear2(g_up, B, H, W, Hd, backward=True)
                _, gyP = gate_split(gy, None, M, want_f32=False)
                gx = rcu_bwd(gy, gyP, sv, base + 4 * n)
                if blk == 3:
                    g_r[3] = gx
                else:
                    g_s = gx            # dL/d s, s = relu(t): gate with the saved post-residual mask next
            else:
                my = sv[3]
                gtF, gtP = gate_split(g_s, my, M)        # dL/dt, t = RCU1(r_blk) + up_prev
                g_r[blk] = rcu_bwd(gtF, gtP, sv, base + 4 * n)
                g_up = gtF                              # d t / d up_prev = identity

        # ---- conv_i (3x3, no bias): gate by relu(f_i) > 0, then dW over the tap tokens
        for i in range(4):
            C, H, W = dims[i]
            M = B * H * W
            _, gfP = gate_split(g_r[i], ctx.mf[i], M, want_f32=False)
            grads[i] = new_like(det[i])
            cv.conv_dw(gfP, Hd, ctx.toks[i], C, cv.geom(B, H, W, C, 3, 3, 1, 1), Hd, grads[i], precision=pr)
        return (None, None, None, None, *grads)


def dpt_res_logits(toks, dims, B, head, precision: int) -> torch.Tensor:
    """Channels-last logits [B, 2*H_0, 2*W_0, K4] of the ResNet-pyramid DPT head before its final nearest x2."""
    return _DPTRes.apply(toks, dims, B, precision, *dpt_res_param_list(head))
