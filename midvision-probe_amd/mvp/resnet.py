"""Frozen ResNet-50 (torchvision v1.5 layout) forward on the HIP conv path
(evals/models/dino_res50.py:38-51,83-101; mocov3_res50.py:97-116).

The trunk is in eval mode in the reference (``dino_resnet.eval()``), so every internal
BatchNorm is folded into its conv at load: w' = w * gamma / sqrt(var + eps), b' = beta - mean * scale.
Activations are channels-last: fp32 [B*H*W, C] where an identity / residual needs them, bf16
pairs as MFMA operands.  stem 7x7/2 = im2col kernel + GEMM; 3x3 (stride 1/2) and the 1x1/2
downsample = implicit-GEMM conv mode; 1x1 = plain GEMM; bottleneck tail = GEMM epilogue
(+identity, ReLU after the add).  Train-mode tap BatchNorm2d = the token BN kernel (tokens = pixels).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import torch

from . import conv as cv
from . import lib, ops, pipeline
from .vit import TapOutputs, parse_precision

LAYERS = (3, 4, 6, 3)


def _fold(sd, conv, bn, eps=1e-5):
    w = sd[conv + ".weight"].float()
    scale = sd[bn + ".weight"].float() / torch.sqrt(sd[bn + ".running_var"].float() + eps)
    return w * scale[:, None, None, None], (sd[bn + ".bias"].float() - sd[bn + ".running_mean"].float() * scale).contiguous()


class ResNetEngine:
    def __init__(self, state_dict: Dict[str, torch.Tensor], *, precision="bf16x3", device="cuda"):
        self.device = torch.device(device)
        self.pr = parse_precision(precision)
        if self.pr == lib.PREC_F16X2:  # a mode of the ViT blocks' GEMMs: the convolution trunk has no two-product kernels
            self.pr = lib.PREC_BF16X3
        sd = {k: v.detach().to(self.device) for k, v in state_dict.items() if v.dtype.is_floating_point}
        pr = self.pr
        # stem: [C0, 3, 7, 7] -> GEMM operand [C0, 147 -> 192]
        w, b = _fold(sd, "conv1", "bn1")
        self.c0 = w.shape[0]
        wk = w.permute(0, 2, 3, 1).reshape(self.c0, -1)  # k = (ky*7 + kx)*3 + c
        self.stem_k = 160  # 7*7*3 = 147 padded to a multiple of 32 (BK = 32 tile)
        wpad = wk.new_zeros(self.c0, self.stem_k)
        wpad[:, : wk.shape[1]] = wk
        self.stem_w, self.stem_b = ops.split_bf16(wpad.contiguous(), pr), b
        self.blocks: List[List[dict]] = []
        for li, n in enumerate(LAYERS, start=1):
            stage = []
            for bi in range(n):
                p = f"layer{li}.{bi}."
                blk = {}
                w1, blk["b1"] = _fold(sd, p + "conv1", p + "bn1")
                blk["w1"] = ops.split_bf16(w1.reshape(w1.shape[0], -1).contiguous(), pr)
                w2, blk["b2"] = _fold(sd, p + "conv2", p + "bn2")
                blk["w2"] = cv.pack_weight(w2, 0, pr)
                w3, blk["b3"] = _fold(sd, p + "conv3", p + "bn3")
                blk["w3"] = ops.split_bf16(w3.reshape(w3.shape[0], -1).contiguous(), pr)
                blk["width"], blk["cout"], blk["cin"] = w1.shape[0], w3.shape[0], w1.shape[1]
                blk["stride"] = 2 if (bi == 0 and li > 1) else 1
                if p + "downsample.0.weight" in sd:
                    wd, blk["bd"] = _fold(sd, p + "downsample.0", p + "downsample.1")
                    blk["wd"] = cv.pack_weight(wd, 0, pr)
                stage.append(blk)
            self.blocks.append(stage)
        self.stage_channels = [self.c0] + [st[-1]["cout"] for st in self.blocks]
        pipeline.publish()  # the folded / split weights are read by forwards on any stream

    # ------------------------------------------------------------------ stages (x = (fp32, pair), H, W)
    def _stem(self, images: torch.Tensor, want_f32: bool = False):
        """conv1 7x7/2 + folded bn1 + ReLU + maxpool 3x3/2 in ONE kernel (csrc/stem.hip): image -> channels-last pooled map.
        (MVP_STEM=im2col selects the older im2col + GEMM + pool form, kept for A/B measurements and as a cross-check.)"""
        B, Cin, H, W = images.shape
        pr, dev = self.pr, self.device
        Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        if Cin == 3 and self.c0 == 64 and os.environ.get("MVP_STEM", "fused") != "im2col":
            Hp, Wp = (Ho + 2 - 3) // 2 + 1, (Wo + 2 - 3) // 2 + 1
            pf = torch.empty(B * Hp * Wp, self.c0, dtype=torch.float32, device=dev) if want_f32 else None
            pp = ops.empty_pair((B * Hp * Wp, self.c0), pr, dev)
            lib.call("mvp_stem7x7_pool", lib.StemArgs(lib.ptr(images), lib.ptr(self.stem_w[0]), lib.ptr(self.stem_w[1]), lib.ptr(self.stem_b), lib.ptr(pf),
                                                       lib.ptr(pp[0]), lib.ptr(pp[1]), B, H, W, pr))
            return pf, pp, Hp, Wp
        col = ops.empty_pair((B * Ho * Wo, self.stem_k), pr, dev)
        lib.call("mvp_im2col_nchw", lib.Im2colArgs(lib.ptr(images), lib.ptr(col[0]), lib.ptr(col[1]), B, Cin, H, W, Ho, Wo, 7, 7, 2, 3, self.stem_k))
        y = torch.empty(B * Ho * Wo, self.c0, dtype=torch.float32, device=dev)
        ops.gemm(col, self.stem_w, B * Ho * Wo, self.c0, self.stem_k, bias=self.stem_b, out_f32=y, act=lib.ACT_RELU, precision=pr)
        Hp, Wp = (Ho + 2 - 3) // 2 + 1, (Wo + 2 - 3) // 2 + 1
        pf = torch.empty(B * Hp * Wp, self.c0, dtype=torch.float32, device=dev)
        pp = ops.empty_pair((B * Hp * Wp, self.c0), pr, dev)
        lib.call("mvp_maxpool_cl", lib.MaxpoolClArgs(lib.ptr(y), lib.ptr(pf), lib.ptr(pp[0]), lib.ptr(pp[1]), B, Ho, Wo, self.c0, Hp, Wp, 3, 2, 1))
        return pf, pp, Hp, Wp

    def _bottleneck(self, blk: dict, xP, B, H, W, want_f32: bool):
        """One bottleneck on channels-last bf16 pairs.  The identity is read back from the block input PAIR (hi + lo,
        2^-17 relative) — an fp32 copy of every block output would add a third of the HBM traffic of layer1; the fp32
        map is produced only for the block whose output is tapped (``want_f32``)."""
        pr, dev = self.pr, self.device
        s = blk["stride"]
        Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
        M_in, M_out = B * H * W, B * Ho * Wo
        o1 = ops.empty_pair((M_in, blk["width"]), pr, dev)
        ops.gemm(xP, blk["w1"], M_in, blk["width"], blk["cin"], bias=blk["b1"], out=o1, act=lib.ACT_RELU, precision=pr)
        o2 = ops.empty_pair((M_out, blk["width"]), pr, dev)
        cv.conv_gemm(o1, cv.geom(B, H, W, blk["width"], 3, 3, s, 1), blk["w2"], blk["width"], bias=blk["b2"], act=lib.ACT_RELU, out=o2, precision=pr)
        if "wd" in blk:
            idt = ops.empty_pair((M_out, blk["cout"]), pr, dev)
            cv.conv_gemm(xP, cv.geom(B, H, W, blk["cin"], 1, 1, s, 0), blk["wd"], blk["cout"], bias=blk["bd"], out=idt, precision=pr)
        else:
            idt = xP
        yF = torch.empty(M_out, blk["cout"], dtype=torch.float32, device=dev) if want_f32 else None
        yP = ops.empty_pair((M_out, blk["cout"]), pr, dev)
        ops.gemm(o2, blk["w3"], M_out, blk["cout"], blk["width"], bias=blk["b3"], residual_pair=idt, out_f32=yF, out=yP,
                 act=lib.ACT_RELU, act_after_res=True, precision=pr)
        return yF, yP, Ho, Wo

    # ------------------------------------------------------------------ forward
    def forward_taps(self, images: torch.Tensor, multilayers: Sequence[int], *, bn: Optional[Sequence[Optional[dict]]] = None,
                     bn_mode: int = 0, want_tokens: bool = False) -> TapOutputs:
        """images: [B,3,S,S] fp32 device (already resized).  Returns NCHW fp32 maps for the stage
        indices in ``multilayers`` (0 = stem+maxpool, 1..4 = layer1..4); ``outs.tokens[j]`` holds the
        channels-last bf16 pair of tap j when ``want_tokens`` (no head consumes it today: off by default, it is a second
        full-size write per tap).  bn[i] is indexed by STAGE i.  Every buffer is allocated on the launching stream, so forwards on
        different streams (mvp/pipeline.py) share nothing but the weights and the tap-BN running statistics (deferred to the consumer)."""
        images = images.to(self.device, torch.float32).contiguous()
        B = images.shape[0]
        outs = TapOutputs()
        outs.tokens, outs.dims = [], []
        xF, xP, H, W = self._stem(images, want_f32=0 in multilayers)
        last = max(multilayers)
        for i in range(5):
            if i > 0:
                stage = self.blocks[i - 1]
                for j, blk in enumerate(stage):
                    xF, xP, H, W = self._bottleneck(blk, xP, B, H, W, want_f32=(i in multilayers) and j == len(stage) - 1)
            if i in multilayers:
                C = self.stage_channels[i]
                HW = H * W
                nchw = torch.empty(B, C, H, W, dtype=torch.float32, device=self.device)
                tok = ops.empty_pair((B * HW, C), self.pr, self.device) if want_tokens else None
                ws = torch.empty(ops.bn_tokens_workspace_bytes(B * HW, C) // 4 + 16, dtype=torch.float32, device=self.device)
                stats = torch.empty(3 * C, dtype=torch.float32, device=self.device)
                b = bn[i] if bn is not None else None
                # the running-statistics update of a forward in flight on a side stream is applied by the consumer, in batch order
                defer = b is not None and bn_mode == 0 and pipeline.pipelined()
                ops.bn_tokens_to_nchw(xF, B, HW, C, HW, workspace=ws, stats=stats,
                                      gamma=b["weight"] if b else None, beta=b["bias"] if b else None,
                                      running_mean=b["running_mean"] if b else None, running_var=b["running_var"] if b else None,
                                      nchw=nchw, tok=tok, ld_tok=C, col_off=0, mode=bn_mode,
                                      num_batches_tracked=b.get("num_batches_tracked") if b else None, defer_running=defer)
                if defer and b.get("running_mean") is not None:
                    pipeline.defer(lambda st=stats, b=b, C=C: ops.bn_running_update(st, b["running_mean"], b["running_var"], b.get("num_batches_tracked"), C))
                outs.append(nchw)
                outs.tokens.append(tok)
                outs.dims.append((C, H, W))
            if i == last:
                break
        return outs
