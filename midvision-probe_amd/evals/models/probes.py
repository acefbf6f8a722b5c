"""evals.models.probes — drop-in for the reference's probe heads (evals/models/probes.py:86-459):
same class names, constructor kwargs, ``.name`` strings and state-dict keys
(``head.conv.weight``, ``head.conv_0.weight``, ``head.ref_0.resConfUnit1.conv.0.weight`` ...),
forward/backward on the HIP kernels (mvp.functional).

nn.Conv2d modules are kept as PARAMETER CONTAINERS only (so reference ``ckpt.pth["probe"]``
state dicts load unchanged); their arithmetic runs in the HIP GEMM / conv kernels.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from mvp import backbone as bb
from mvp import functional as MF
from mvp import lib
from mvp.vit import parse_precision


def _precision(p):
    pr = parse_precision(p or bb.default_precision())
    # 'f16x2' is a mode of the frozen ViT blocks' GEMMs (two products against frozen weights); the trained probe keeps three
    from mvp import lib

    return lib.PREC_BF16X3 if pr == lib.PREC_F16X2 else pr


class SurfaceNormalHead(nn.Module):
    """Reference: probes.py:86-116."""

    def __init__(self, feat_dim, head_type="multiscale", uncertainty_aware=False, hidden_dim=512, kernel_size=1, precision=None):
        super().__init__()
        self.uncertainty_aware = uncertainty_aware
        output_dim = 4 if uncertainty_aware else 3
        self.kernel_size = kernel_size
        assert head_type in ["linear", "multiscale", "dpt"]
        name = f"snorm_{head_type}_k{kernel_size}"
        self.name = f"{name}_UA" if uncertainty_aware else name
        if head_type == "linear":
            self.head = Linear(feat_dim, output_dim, kernel_size, precision=precision)
        elif head_type == "dpt":
            self.head = DPT(feat_dim, output_dim, hidden_dim, kernel_size, precision=precision)
        else:
            self.head = MultiscaleHead(feat_dim, output_dim, hidden_dim, kernel_size, precision=precision)

    def forward(self, feats):
        return self.head(feats)


class DepthHead(nn.Module):
    """Reference: probes.py:119-157."""

    def __init__(self, feat_dim, head_type="multiscale", min_depth=0.001, max_depth=10, prediction_type="sigdepth", hidden_dim=512,
                 kernel_size=1, precision=None):
        super().__init__()
        self.kernel_size = kernel_size
        self.name = f"{prediction_type}_{head_type}_k{kernel_size}"
        if prediction_type == "bindepth":
            output_dim = 256
            self.predict = DepthBinPrediction(min_depth, max_depth, n_bins=output_dim)
        elif prediction_type == "sigdepth":
            output_dim = 1
            self.predict = DepthSigmoidPrediction(min_depth, max_depth)
        else:
            raise ValueError()
        if head_type == "linear":
            self.head = Linear(feat_dim, output_dim, kernel_size, precision=precision)
        elif head_type == "dpt":
            self.head = DPT(feat_dim, output_dim, hidden_dim, kernel_size, precision=precision)
        else:
            self.head = MultiscaleHead(feat_dim, output_dim, hidden_dim, kernel_size, precision=precision)

    def forward(self, feats):
        """Prediction each pixel."""
        if isinstance(self.head, DPT):
            # the final nearest x2 of DPT commutes with the per-pixel predictor: predict at 8h x 8w,
            # then upsample the 1-channel depth (instead of 256 channels at full resolution)
            depth = self.predict(self.head(feats, defer_upsample=True))
            return MF.interpolate(depth, scale_factor=2, mode="nearest")
        if (isinstance(self.head, Linear) and self.head.kernel_size == 1 and isinstance(self.predict, DepthBinPrediction)
                and self.head.conv.out_channels % 8 == 0):
            # fused GEMM -> [bilinear x4 + bin expectation]: same arithmetic as head -> predict, without
            # materialising the x4-upsampled logits (mvp/functional.py:_LinearBinsHead)
            if type(feats) is not list and not isinstance(feats, (list, tuple)):
                feats = [feats]
            return MF.linear_bins_head(feats, self.head.conv.weight, self.head.conv.bias, self.head.precision,
                                       self.predict.n_bins, self.predict.min_depth, self.predict.max_depth)
        feats = self.head(feats)
        return self.predict(feats)


def _as_channels_last(x: torch.Tensor) -> torch.Tensor:
    """[B,K,H,W] (any strides) -> contiguous [B,H,W,K] without a copy when x is already a
    permuted view of a channels-last buffer (what Linear/DPT return)."""
    y = x.permute(0, 2, 3, 1)
    return y if y.is_contiguous() else y.contiguous()


class DepthBinPrediction(nn.Module):
    """Reference: probes.py:160-200 ('UD' bins, 'linear' normalisation)."""

    def __init__(self, min_depth=0.001, max_depth=10, n_bins=256, bins_strategy="UD", norm_strategy="linear"):
        super().__init__()
        if bins_strategy != "UD" or norm_strategy != "linear":
            raise NotImplementedError("only the reference defaults (UD bins, linear norm) are on the hot path")
        self.n_bins, self.min_depth, self.max_depth = n_bins, min_depth, max_depth
        self.norm_strategy, self.bins_strategy = norm_strategy, bins_strategy

    def forward(self, prob):
        return MF.depth_bins(_as_channels_last(prob), self.n_bins, self.min_depth, self.max_depth)


class DepthSigmoidPrediction(nn.Module):
    """Reference: probes.py:203-212."""

    def __init__(self, min_depth=0.001, max_depth=10):
        super().__init__()
        self.min_depth, self.max_depth = min_depth, max_depth

    def forward(self, pred):
        return MF.depth_sigmoid(_as_channels_last(pred), self.min_depth, self.max_depth)


class Linear(nn.Module):
    """Reference: probes.py:417-432 — cat maps, bilinear x4, conv k x k.  For k = 1 the conv
    and the bilinear resample commute, so the GEMM runs at token resolution (16x fewer rows)."""

    def __init__(self, input_dim, output_dim, kernel_size=1, precision=None):
        super().__init__()
        if type(input_dim) is not int:
            input_dim = sum(input_dim)
        assert type(input_dim) is int
        padding = kernel_size // 2
        self.conv = nn.Conv2d(input_dim, output_dim, kernel_size, padding=padding)
        self.kernel_size = kernel_size
        self.precision = _precision(precision)

    def forward(self, feats):
        if type(feats) is not list and not isinstance(feats, (list, tuple)):
            feats = [feats]
        K = self.conv.out_channels
        if self.kernel_size != 1:
            lq = MF.linear_head_kxk(feats, self.conv.weight, self.conv.bias, self.precision)
        else:
            lq = MF.linear_head_k1(feats, self.conv.weight, self.conv.bias, self.precision)  # [B,4h,4w,K4]
        return lq[..., :K].permute(0, 3, 1, 2)  # NCHW view of the channels-last logits


def make_conv(input_dim, hidden_dim, output_dim, num_layers, kernel_size=1):
    """Reference: probes.py:400-412 (parameter containers; note: no padding)."""
    if num_layers == 1:
        return nn.Conv2d(input_dim, output_dim, kernel_size)
    assert num_layers > 1
    modules = [nn.Conv2d(input_dim, hidden_dim, kernel_size), nn.ReLU(inplace=True)]
    for _ in range(num_layers - 2):
        modules += [nn.Conv2d(hidden_dim, hidden_dim, kernel_size), nn.ReLU(inplace=True)]
    modules.append(nn.Conv2d(hidden_dim, output_dim, kernel_size))
    return nn.Sequential(*modules)


class MultiscaleHead(nn.Module):
    """Reference: probes.py:435-458 (DepthHead's / SurfaceNormalHead's default head_type).  kernel_size 1 (the default) runs on the
    HIP path with the 1x1 convs commuted below the resamples (mvp/multiscale.py).  kernel_size > 1: the reference's convs carry no
    padding (probes.py:400-412), so every conv shrinks its map by k - 1 and nothing commutes; the graph is evaluated as written, each
    conv on the implicit-GEMM kernels (mvp.functional.conv2d_valid), the resamples on the resize kernel."""

    def __init__(self, input_dims, output_dim, hidden_dim=512, kernel_size=1, precision=None):
        super().__init__()
        input_dims = [d if isinstance(d, int) else d[0] for d in input_dims]
        self.convs = nn.ModuleList([make_conv(in_d, None, hidden_dim, 1, kernel_size) for in_d in input_dims])
        interm_dim = len(input_dims) * hidden_dim
        self.conv_mid = make_conv(interm_dim, hidden_dim, hidden_dim, 3, kernel_size)
        self.conv_out = make_conv(hidden_dim, hidden_dim, output_dim, 2, kernel_size)
        self.kernel_size, self.input_dims = kernel_size, input_dims
        self.precision = _precision(precision)

    def forward(self, feats):
        from mvp import multiscale as ms

        if self.kernel_size != 1:
            return self._forward_kxk(list(feats))
        feats = list(feats)
        h, w = feats[-1].shape[-2:]
        # bilinear resample to the last map's size (probes.py:449); it commutes with the 1x1 conv that precedes it in the reference
        feats = [f if tuple(f.shape[-2:]) == (h, w) else MF.interpolate(f, size=(h, w), mode="bilinear") for f in feats]
        pack = MF.pack_features(feats, self.precision)
        lq = ms.multiscale_logits(pack, self.input_dims, self, self.precision)
        K = self.conv_out[2].out_channels
        return lq[..., :K].permute(0, 3, 1, 2)


def _multiscale_forward_kxk(self, feats):
    """probes.py:447-458 as written (un-padded k x k convs).  The features are frozen (no gradient into them)."""
    pr = self.precision
    conv = lambda m, x, dx=True: MF.conv2d_valid(x, m.weight, m.bias, pr, need_input_grad=dx)  # noqa: E731
    fs = [conv(self.convs[i], f.detach(), False) for i, f in enumerate(feats)]
    h, w = fs[-1].shape[-2:]
    fs = [f if tuple(f.shape[-2:]) == (h, w) else MF.interpolate(f, size=(h, w), mode="bilinear") for f in fs]
    x = torch.cat(fs, dim=1).relu()
    x = MF.interpolate(x, scale_factor=2, mode="bilinear")
    x = conv(self.conv_mid[0], x).relu()
    x = conv(self.conv_mid[2], x).relu()
    x = conv(self.conv_mid[4], x).relu()
    x = MF.interpolate(x, scale_factor=4, mode="bilinear")
    x = conv(self.conv_out[0], x).relu()
    return conv(self.conv_out[2], x)


MultiscaleHead._forward_kxk = _multiscale_forward_kxk


class DPT(nn.Module):
    """Reference: probes.py:309-399.  Transformer variant (4 equal-resolution taps): mvp/dpt.py; ResNet-pyramid variant
    (4 maps at 8x/4x/2x/1x resolution, bias-free 3x3 input convs, pre-activation fusion units): mvp/dpt_res.py."""

    def __init__(self, input_dims, output_dim, hidden_dim=512, kernel_size=3, precision=None):
        super().__init__()
        assert len(input_dims) == 4
        self.resnet = not isinstance(input_dims[0], int)
        for i in range(4):
            if self.resnet:
                conv = nn.Conv2d(input_dims[i][0], hidden_dim, kernel_size=3, stride=1, padding=1, bias=False)
            else:
                conv = nn.Conv2d(input_dims[i], hidden_dim, 1, padding=0)
            setattr(self, f"conv_{i}", conv)
        for i in range(4):
            setattr(self, f"ref_{i}", FeatureFusionBlock(hidden_dim, kernel_size, is_transformer=not self.resnet, with_skip=(i != 3)))
        self.out_conv = nn.Sequential(nn.Conv2d(hidden_dim, hidden_dim, 3, padding=1), nn.ReLU(True), nn.Conv2d(hidden_dim, output_dim, 3, padding=1))
        self.precision = _precision(precision)

    def forward(self, feats, defer_upsample=False):
        """Prediction each pixel.  ``defer_upsample=True`` (used by DepthHead) returns the
        logits before the final nearest x2 (which commutes with the per-pixel predictor)."""
        assert len(feats) == 4
        if self.resnet:
            from mvp import dpt_res, ops

            B = feats[0].shape[0]
            toks, dims = [], []
            for f in feats:  # NCHW fp32 pyramid -> channels-last bf16 pairs (operands of the 3x3 input convs)
                if not f.is_cuda:
                    raise lib.MvpError("probe features must be device tensors (no CPU fallback)")
                _, C, H, W = f.shape
                tok = ops.empty_pair((B * H * W, C), self.precision, f.device)
                ops.pack_nchw_tokens(f.contiguous().float(), B, C, H * W, tok=tok, ld_tok=C, col_off=0)
                toks.append(tok)
                dims.append((C, H, W))
            lq = dpt_res.dpt_res_logits(toks, dims, B, self, self.precision)  # [B, 2*H_0, 2*W_0, K4]
        else:
            from mvp import dpt as mdpt

            pack = MF.pack_features(list(feats), self.precision)
            lq = mdpt.dpt_vit_logits(pack, self, self.precision)  # [B, 8h, 8w, K4] channels-last
        K = self.out_conv[2].out_channels
        y = lq[..., :K].permute(0, 3, 1, 2)
        if defer_upsample:
            return y
        return MF.interpolate(y.contiguous(), scale_factor=2, mode="nearest")


class FeatureFusionBlock(nn.Module):
    """Reference: probes.py:215-259 (parameter container)."""

    def __init__(self, features, kernel_size=3, with_skip=True, upsample=False, is_transformer=False):
        super().__init__()
        self.with_skip, self.upsample, self.is_transformer = with_skip, upsample, is_transformer
        if self.with_skip:
            self.resConfUnit1 = ResidualConvUnit(features, kernel_size, is_transformer=is_transformer)
        self.resConfUnit2 = ResidualConvUnit(features, kernel_size, is_transformer=is_transformer)


class ResidualConvUnit(nn.Module):
    """Reference: probes.py:262-306 (parameter container)."""

    def __init__(self, features, kernel_size=3, is_transformer=False, inplace_relu=True):
        super().__init__()
        if is_transformer:
            assert kernel_size % 2 == 1, "Kernel size needs to be odd for transformer-based implementation"
            padding = kernel_size // 2
            self.conv = nn.Sequential(nn.Conv2d(features, features, kernel_size, padding=padding), nn.ReLU(inplace=inplace_relu),
                                      nn.Conv2d(features, features, kernel_size, padding=padding), nn.ReLU(inplace=inplace_relu))
        else:
            self.conv1 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
            self.conv2 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
            self.relu = nn.ReLU(inplace=inplace_relu)
