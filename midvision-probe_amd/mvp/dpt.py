"""DPT probe head (evals/models/probes.py:215-399, transformer variant) on the HIP conv path.

Everything is channels-last: activations are [B*H*W, C] fp32 (residual / skip operands) plus
bf16 pairs (MFMA operands); 3x3 convs are implicit GEMMs (gemm.hip conv mode) with fused
bias + ReLU + skip adds + byte-mask output; nearest upsamples are folded into the consumer's
addressing where the tensor would be large (x4 before out_conv).  The backward pass is written
out by hand (one autograd.Function for the whole head): data gradients reuse the conv GEMM with
flipped/transposed weights and a fused ReLU gate, weight gradients use the TN split-K kernel.

Reference graph (probes.py:377-399):
  f_i = conv_i(feat_i) (1x1)            -> nearest x2
  out = ref_3(f_3); ref_2(f_2, out); ref_1(f_1, out); ref_0(f_0, out)
        ref(x, skip): x = RCU1(x) + skip (if skip);  x = RCU2(x);   RCU(x) = relu(conv(relu(conv(x)))) + x
  out -> nearest x4 -> out_conv.0 (3x3) -> ReLU -> out_conv.2 (3x3) -> nearest x2
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import conv as cv
from . import lib, ops
from .lib import ACT_NONE, ACT_RELU
from .vit import PackedFeatures

RCU_ORDER = [(3, 2), (2, 1), (2, 2), (1, 1), (1, 2), (0, 1), (0, 2)]  # (ref block, unit) in execution order


def dpt_param_list(head) -> List[torch.Tensor]:
    """Flatten the DPT module's parameters in the order _DPTViT expects."""
    ps = []
    for i in range(4):
        c = getattr(head, f"conv_{i}")
        ps += [c.weight, c.bias]
    for blk, unit in RCU_ORDER:
        rcu = getattr(getattr(head, f"ref_{blk}"), f"resConfUnit{unit}")
        ps += [rcu.conv[0].weight, rcu.conv[0].bias, rcu.conv[2].weight, rcu.conv[2].bias]
    ps += [head.out_conv[0].weight, head.out_conv[0].bias, head.out_conv[2].weight, head.out_conv[2].bias]
    return ps


def _up(n, m):
    return (n + m - 1) // m * m


class _DPTViT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pack: PackedFeatures, precision: int, *params):
        pr = precision
        dev = params[0].device
        B, h, w = pack.B, pack.h, pack.w
        C = pack.Ctot // 4
        Hd = params[0].shape[0]
        k = params[8].shape[-1]
        Cout = params[-2].shape[0]
        if C % 128 or Hd % 128:
            raise lib.MvpError(f"DPT on the HIP path needs feature and hidden channels that are multiples of 128 (got {C}, {Hd})")
        K4 = _up(Cout, 4)
        M0, H1, W1 = B * h * w, 2 * h, 2 * w
        M1, H2, W2 = B * H1 * W1, 8 * h, 8 * w
        M2 = B * H2 * W2
        det = [p.detach() for p in params]

        # ---- conv_i (1x1) at token resolution on the packed features, then nearest x2
        u = []
        for i in range(4):
            wi = ops.split_bf16(det[2 * i].reshape(Hd, C).float().contiguous(), pr)
            f = torch.empty(M0, Hd, dtype=torch.float32, device=dev)
            a = (pack.tok[0][:, i * C:], pack.tok[1][:, i * C:] if pack.tok[1] is not None else None)
            ops.gemm(a, wi, M0, Hd, C, bias=det[2 * i + 1].float().contiguous(), out_f32=f, precision=pr, lda=pack.Cpad)
            # (fp32, pair) at H1 x W1, and the COARSE pair: a 3x3 unit's first conv reads the x2 map through its coarse tap products
            u.append(cv.upsample_nearest(f, B, h, w, Hd, 2, precision=pr, want_pair=(k != 3)) + (ops.split_bf16(f, pr) if k == 3 else None,))

        g1 = cv.geom(B, H1, W1, Hd, k, k, 1, k // 2)
        saved_rcu = []

        def rcu(xF, xP, wa, ba, wb, bb, extra=None, coarse=None):
            """``coarse``: the unit's input is the nearest x2 of this [M0, Hd] pair (u_blk): its first 3x3 conv then runs from the coarse
            grid's tap products (cv.upconv3_forward) and its backward folds onto the coarse grid (rcu_bwd)."""
            aP = ops.empty_pair((M1, Hd), pr, dev)
            ma = torch.empty(M1, Hd, dtype=torch.uint8, device=dev)
            if coarse is not None:
                cv.upconv3_forward(coarse, wa, ba.float().contiguous(), B, h, w, 2, act=ACT_RELU, out=aP, out_mask=ma, precision=pr)
            else:
                cv.conv_gemm(xP, g1, cv.pack_weight(wa, 0, pr), Hd, bias=ba.float().contiguous(), act=ACT_RELU, out=aP, out_mask=ma, precision=pr)
            yF = torch.empty(M1, Hd, dtype=torch.float32, device=dev)
            yP = ops.empty_pair((M1, Hd), pr, dev)
            mb = torch.empty(M1, Hd, dtype=torch.uint8, device=dev)
            cv.conv_gemm(aP, g1, cv.pack_weight(wb, 0, pr), Hd, bias=bb.float().contiguous(), act=ACT_RELU, residual=xF, residual2=extra,
                         out_f32=yF, out=yP, out_mask=mb, precision=pr)
            saved_rcu.append((xP, aP, ma, mb, coarse))
            return yF, yP

        base = 8
        out = None
        for n, (blk, unit) in enumerate(RCU_ORDER):
            wa, ba, wb, bb = det[base + 4 * n: base + 4 * n + 4]
            if unit == 1:      # x = RCU1(f_blk) + out
                out = rcu(u[blk][0], u[blk][1], wa, ba, wb, bb, extra=out[0], coarse=u[blk][2])
            elif blk == 3:     # ref_3 has no skip unit: RCU2 acts on f_3
                out = rcu(u[3][0], u[3][1], wa, ba, wb, bb, coarse=u[3][2])
            else:
                out = rcu(out[0], out[1], wa, ba, wb, bb)
        o0P = out[1]

        # ---- out_conv: the x4 nearest upsample + first 3x3 evaluated from the coarse grid's per-tap products (cv.upconv3_forward: one
        # GEMM over M1 coarse pixels with N = 9*Hd, then a gather-sum per fine pixel) instead of a convolution over 16x as many pixels
        w0, b0, w2, b2 = det[-4:]
        h0P = ops.empty_pair((M2, Hd), pr, dev)
        m0 = torch.empty(M2, Hd, dtype=torch.uint8, device=dev)
        cv.upconv3_forward(o0P, w0, b0.float().contiguous(), B, H1, W1, 4, act=ACT_RELU, out=h0P, out_mask=m0, precision=pr)
        g3 = cv.geom(B, H2, W2, Hd, 3, 3, 1, 1)
        b2p = torch.cat([b2.float(), b2.new_zeros(K4 - Cout).float()]) if K4 != Cout else b2.float().contiguous()
        logits = torch.empty(B, H2, W2, K4, dtype=torch.float32, device=dev)
        cv.conv_gemm(h0P, g3, cv.pack_weight(w2, 0, pr, pad_cout_to=K4), K4, bias=b2p, out_f32=logits, precision=pr)

        ctx.pack, ctx.pr, ctx.dims = pack, pr, (B, h, w, C, Hd, k, Cout, K4)
        ctx.saved_rcu, ctx.o0P, ctx.h0P, ctx.m0 = saved_rcu, o0P, h0P, m0
        ctx.generation = pack.generation
        ctx.param_refs = params  # the Parameter objects themselves: their FlatAdamW gradient slots are looked up in backward
        ctx.save_for_backward(*params)
        return logits

    @staticmethod
    def backward(ctx, g_logits):
        pack, pr = ctx.pack, ctx.pr
        B, h, w, C, Hd, k, Cout, K4 = ctx.dims
        params = ctx.saved_tensors
        det = [p.detach() for p in params]
        dev = g_logits.device
        if pack.generation != ctx.generation:
            raise lib.MvpError("DPT backward: the backbone ran again before this backward and overwrote this step's packed features")
        M0, H1, W1 = B * h * w, 2 * h, 2 * w
        M1, H2, W2 = B * H1 * W1, 8 * h, 8 * w
        M2 = B * H2 * W2
        grads: List[Optional[torch.Tensor]] = [None] * len(params)

        from .functional import _grad_dst

        dst_of = {id(d): r for d, r in zip(det, getattr(ctx, "param_refs", det))}

        def new_like(p):
            """Gradient buffer for parameter ``p``: its slot in FlatAdamW's flat gradient (autograd then adopts it: no copy, no
            accumulate-add), else a fresh tensor."""
            d = _grad_dst(dst_of.get(id(p)), tuple(p.shape))
            return d if d is not None else torch.empty(p.shape, dtype=torch.float32, device=dev)

        def bias_grad(gF, N, n_true=None):
            db = torch.empty(N, dtype=torch.float32, device=dev)
            ops.colsum(gF, db, gF.shape[0], N)
            return db if n_true is None else db[:n_true].contiguous()

        # ---- out_conv.2
        w0, b0, w2, b2 = det[-4:]
        g_logits = g_logits.contiguous().float().reshape(M2, K4)
        LG = _up(K4, 128)
        gP = cv.mask_split(g_logits, None, M2, K4, ldo=LG, precision=pr)
        g3 = cv.geom(B, H2, W2, Hd, 3, 3, 1, 1)
        grads[-2] = new_like(w2)
        cv.conv_dw(gP, LG, ctx.h0P, Hd, g3, Cout, grads[-2], precision=pr)
        grads[-1] = bias_grad(g_logits, K4, Cout)
        gh0F = torch.empty(M2, Hd, dtype=torch.float32, device=dev)  # (fp32 only: its consumers are the box sums and the bias gradient)
        gd = cv.geom(B, H2, W2, LG, 3, 3, 1, 1)
        cv.conv_gemm(gP, gd, cv.pack_weight(w2, 1, pr, pad_cout_to=LG), Hd, relu_mask=ctx.m0, mask_mode=2, out_f32=gh0F, precision=pr)
        # ---- out_conv.0 (input = x4 nearest of o0): both gradients folded onto the COARSE grid.  The upsampled input is constant over
        # 4x4 blocks, so with G = the 4x4 box sums of the output gradient per coarse pixel and tap (one pass over gh0, csrc/conv.hip
        # upconv3_boxsum_kernel), dW = Gᵀ·o0 and d(o0) = G·Wᵀ are GEMMs over the M1 coarse pixels with K = 9·Hd — 16x fewer MFMA
        # flops than the weight- and input-gradient convolutions over the 16·M1 fine pixels (+ the adjoint of the upsample) they replace.
        GP = cv.upconv3_grad_boxsum(gh0F, B, H1, W1, Hd, 4, precision=pr)  # [M1, 9*Hd], column tap*Hd + co
        dw9 = torch.empty(9 * Hd, Hd, 1, 1, dtype=torch.float32, device=dev)
        cv.conv_dw(GP, 9 * Hd, ctx.o0P, Hd, cv.geom(B, H1, W1, Hd, 1, 1, 1, 0), 9 * Hd, dw9, precision=pr)
        grads[-4] = new_like(w0)
        grads[-4].copy_(dw9.view(3, 3, Hd, Hd).permute(2, 3, 0, 1))  # [(ky,kx), co, ci] -> [co, ci, ky, kx]
        grads[-3] = bias_grad(gh0F, Hd)
        wd = ops.split_bf16(w0.float().permute(1, 2, 3, 0).reshape(Hd, 9 * Hd).contiguous(), pr)  # [ci, (ky, kx, co)]
        gy = torch.empty(M1, Hd, dtype=torch.float32, device=dev)
        ops.gemm(GP, wd, M1, Hd, 9 * Hd, out_f32=gy, precision=pr)
        del GP, gh0F

        g1 = cv.geom(B, H1, W1, Hd, k, k, 1, k // 2)

        def rcu_bwd(gy, saved, idx):
            xP, aP, ma, mb, coarse = saved
            wa, ba, wb, bb = det[idx: idx + 4]
            grF = torch.empty(M1, Hd, dtype=torch.float32, device=dev)
            grP = ops.empty_pair((M1, Hd), pr, dev)
            a = lib.MaskSplitArgs(lib.ptr(gy), lib.ptr(mb), lib.ptr(grF), lib.ptr(grP[0]), lib.ptr(grP[1]), M1, Hd, Hd, Hd, Hd)
            lib.call("mvp_mask_split", a)
            grads[idx + 2] = new_like(wb)
            cv.conv_dw(grP, Hd, aP, Hd, g1, Hd, grads[idx + 2], precision=pr)
            grads[idx + 3] = bias_grad(grF, Hd)
            gaF = torch.empty(M1, Hd, dtype=torch.float32, device=dev)
            gaP = ops.empty_pair((M1, Hd), pr, dev) if coarse is None else None  # (the coarse path reads the fp32 gradient only)
            cv.conv_gemm(grP, g1, cv.pack_weight(wb, 1, pr), Hd, relu_mask=ma, mask_mode=2, out_f32=gaF, out=gaP, precision=pr)
            grads[idx] = new_like(wa)
            grads[idx + 1] = bias_grad(gaF, Hd)
            if coarse is not None:
                # input = nearest x2 of a coarse map: both gradients on the coarse grid (box sums of ga per coarse pixel and tap), and the
                # unit's input gradient is only ever needed summed over the 2x2 blocks (adjoint of the upsample): gf = G·Wᵀ + blocksum(gy)
                GP = cv.upconv3_grad_boxsum(gaF, B, h, w, Hd, 2, precision=pr)
                dw9 = torch.empty(9 * Hd, Hd, 1, 1, dtype=torch.float32, device=dev)
                cv.conv_dw(GP, 9 * Hd, coarse, Hd, cv.geom(B, h, w, Hd, 1, 1, 1, 0), 9 * Hd, dw9, precision=pr)
                grads[idx].copy_(dw9.view(3, 3, Hd, Hd).permute(2, 3, 0, 1))
                gyc, _ = cv.upsample_nearest(gy, B, h, w, Hd, 2, want_pair=False, precision=pr, backward=True)
                wd = ops.split_bf16(wa.float().permute(1, 2, 3, 0).reshape(Hd, 9 * Hd).contiguous(), pr)
                gf = torch.empty(M0, Hd, dtype=torch.float32, device=dev)
                ops.gemm(GP, wd, M0, Hd, 9 * Hd, residual=gyc, out_f32=gf, precision=pr)
                return gf, True
            cv.conv_dw(gaP, Hd, xP, Hd, g1, Hd, grads[idx], precision=pr)
            gx = torch.empty(M1, Hd, dtype=torch.float32, device=dev)
            cv.conv_gemm(gaP, g1, cv.pack_weight(wa, 1, pr), Hd, residual=gy, out_f32=gx, precision=pr)
            return gx, False

        g_u: List[Optional[torch.Tensor]] = [None] * 4
        base = 8
        for n in reversed(range(len(RCU_ORDER))):
            blk, unit = RCU_ORDER[n]
            gx, on_coarse = rcu_bwd(gy, ctx.saved_rcu[n], base + 4 * n)
            if unit == 1:
                g_u[blk] = (gx, on_coarse)  # gradient of the upsampled f_blk (or already of f_blk); the "+ out" branch receives gy unchanged
            elif blk == 3:
                g_u[3] = (gx, on_coarse)
            else:
                gy = gx            # RCU2's input is the fusion sum x = RCU1(f) + out: both addends get gx
                continue
            # unit == 1: gy (= gradient wrt the fusion sum) flows on to the previous block's output
        # NB: for unit==1 the loop leaves gy untouched, which is exactly d(sum)/d(out) = identity.

        # ---- conv_i (1x1): adjoint of nearest x2, then dW over the packed feature tokens
        g0 = cv.geom(B, h, w, C, 1, 1, 1, 0)
        for i in range(4):
            if g_u[i][1]:
                gfF, gfP = g_u[i][0], ops.split_bf16(g_u[i][0], pr)
            else:
                gfF, gfP = cv.upsample_nearest(g_u[i][0], B, h, w, Hd, 2, precision=pr, backward=True)
            grads[2 * i] = new_like(det[2 * i])
            x = (pack.tok[0][:, i * C:], pack.tok[1][:, i * C:] if pack.tok[1] is not None else None)
            cv.conv_dw(gfP, Hd, x, pack.Cpad, g0, Hd, grads[2 * i], precision=pr)
            grads[2 * i + 1] = bias_grad(gfF, Hd)
        return (None, None, *grads)


def dpt_vit_logits(pack: PackedFeatures, head, precision: int) -> torch.Tensor:
    """Channels-last logits [B, 8h, 8w, K4] of the DPT head BEFORE its final nearest x2."""
    return _DPTViT.apply(pack, precision, *dpt_param_list(head))
