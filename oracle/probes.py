"""Oracle: probe heads (linear / DPT) and depth predictors, fp32 CPU, functional over the
reference's state-dict key layout (``head.conv.weight``, ``head.conv_0.weight``,
``head.ref_0.resConfUnit1.conv.0.weight`` ...).

Test infrastructure only (see oracle/__init__.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

StateDict = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------- init
def _conv_init(g, cout, cin, k, bias=True):
    """torch ``nn.Conv2d`` default init (kaiming-uniform a=sqrt(5)) from a private generator."""
    fan_in = cin * k * k
    bound = 1.0 / math.sqrt(fan_in)
    w = (torch.rand(cout, cin, k, k, generator=g) * 2 - 1) * bound
    b = (torch.rand(cout, generator=g) * 2 - 1) * bound if bias else None
    return w, b


def make_linear_head_weights(feat_dims: Sequence[int], out_dim: int, k: int = 1, seed: int = 0) -> StateDict:
    g = torch.Generator().manual_seed(seed)
    w, b = _conv_init(g, out_dim, int(sum(feat_dims)), k)
    return {"head.conv.weight": w, "head.conv.bias": b}


def make_dpt_weights(input_dims, out_dim: int, hidden: int = 512, k: int = 3, seed: int = 0) -> StateDict:
    """Key layout of probes.py:309-375."""
    g = torch.Generator().manual_seed(seed)
    resnet = not isinstance(input_dims[0], int)
    sd: StateDict = {}
    for i in range(4):
        if resnet:
            w, _ = _conv_init(g, hidden, input_dims[i][0], 3, bias=False)
            sd[f"head.conv_{i}.weight"] = w
        else:
            w, b = _conv_init(g, hidden, input_dims[i], 1)
            sd[f"head.conv_{i}.weight"], sd[f"head.conv_{i}.bias"] = w, b
    for i in range(4):
        units = ["resConfUnit2"] if i == 3 else ["resConfUnit1", "resConfUnit2"]
        for u in units:
            if resnet:
                for c in ("conv1", "conv2"):
                    w, b = _conv_init(g, hidden, hidden, 3)
                    sd[f"head.ref_{i}.{u}.{c}.weight"], sd[f"head.ref_{i}.{u}.{c}.bias"] = w, b
            else:
                for c in (0, 2):
                    w, b = _conv_init(g, hidden, hidden, k)
                    sd[f"head.ref_{i}.{u}.conv.{c}.weight"], sd[f"head.ref_{i}.{u}.conv.{c}.bias"] = w, b
    w, b = _conv_init(g, hidden, hidden, 3)
    sd["head.out_conv.0.weight"], sd["head.out_conv.0.bias"] = w, b
    w, b = _conv_init(g, out_dim, hidden, 3)
    sd["head.out_conv.2.weight"], sd["head.out_conv.2.bias"] = w, b
    return sd


def make_multiscale_weights(input_dims, out_dim: int, hidden: int = 512, k: int = 1, seed: int = 0) -> StateDict:
    """Key layout of MultiscaleHead (probes.py:435-445): head.convs.i, head.conv_mid.{0,2,4}, head.conv_out.{0,2}."""
    g = torch.Generator().manual_seed(seed)
    dims = [d if isinstance(d, int) else d[0] for d in input_dims]
    sd: StateDict = {}

    def put(name, cout, cin):
        sd[name + ".weight"], sd[name + ".bias"] = _conv_init(g, cout, cin, k)

    for i, d in enumerate(dims):
        put(f"head.convs.{i}", hidden, d)
    put("head.conv_mid.0", hidden, len(dims) * hidden)
    put("head.conv_mid.2", hidden, hidden)
    put("head.conv_mid.4", hidden, hidden)
    put("head.conv_out.0", hidden, hidden)
    put("head.conv_out.2", out_dim, hidden)
    return sd


# --------------------------------------------------------------------------- heads
def multiscale_head(sd: StateDict, feats: Sequence[torch.Tensor]) -> torch.Tensor:
    """probes.py:447-458 MultiscaleHead.forward (convs carry no padding, probes.py:400-412)."""
    c = lambda n, x: F.conv2d(x, sd[f"head.{n}.weight"], sd[f"head.{n}.bias"])  # noqa: E731
    fs = [c(f"convs.{i}", f) for i, f in enumerate(feats)]
    h, w = fs[-1].shape[-2:]
    x = torch.cat([F.interpolate(f, (h, w), mode="bilinear") for f in fs], dim=1).relu()
    x = F.interpolate(x, scale_factor=2, mode="bilinear")
    x = c("conv_mid.4", c("conv_mid.2", c("conv_mid.0", x).relu()).relu()).relu()
    x = F.interpolate(x, scale_factor=4, mode="bilinear")
    return c("conv_out.2", c("conv_out.0", x).relu())



def linear_head(sd: StateDict, feats, k: int = 1) -> torch.Tensor:
    """probes.py:427-432 — cat maps on C, bilinear x4 (align_corners=False), conv k x k."""
    if isinstance(feats, (list, tuple)):
        feats = torch.cat(list(feats), dim=1)
    feats = F.interpolate(feats, scale_factor=4, mode="bilinear")
    return F.conv2d(feats, sd["head.conv.weight"], sd["head.conv.bias"], padding=k // 2)


def _rcu(sd: StateDict, p: str, x: torch.Tensor, transformer: bool, k: int) -> torch.Tensor:
    """probes.py:262-306 ResidualConvUnit."""
    if transformer:
        y = F.conv2d(x, sd[p + "conv.0.weight"], sd[p + "conv.0.bias"], padding=k // 2).relu()
        y = F.conv2d(y, sd[p + "conv.2.weight"], sd[p + "conv.2.bias"], padding=k // 2).relu()
        return y + x
    # CNN flavour: pre-activation.  The reference's first ReLU is in-place on ``x``
    # (probes.py:275,302), so the skip connection adds relu(x), not x.
    xr = x.relu()
    y = F.conv2d(xr, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1).relu()
    y = F.conv2d(y, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    return y + xr


def _ffb(sd, p, x, skip, transformer, k, with_skip=True):
    """probes.py:244-259 FeatureFusionBlock."""
    if skip is not None and with_skip:
        assert skip.shape == x.shape, "Shape of skip_x must match x"
        x = _rcu(sd, p + "resConfUnit1.", x, transformer, k) + skip
    x = _rcu(sd, p + "resConfUnit2.", x, transformer, k)
    if not transformer:
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    return x


def dpt_head(sd: StateDict, feats: Sequence[torch.Tensor], k: int = 3) -> torch.Tensor:
    """probes.py:377-399 DPT.forward."""
    assert len(feats) == 4
    resnet = "head.conv_0.bias" not in sd
    f = []
    for i in range(4):
        if resnet:
            f.append(F.conv2d(feats[i], sd[f"head.conv_{i}.weight"], None, padding=1))
        else:
            f.append(F.conv2d(feats[i], sd[f"head.conv_{i}.weight"], sd[f"head.conv_{i}.bias"]))
    if not resnet:
        f = [F.interpolate(x, scale_factor=2) for x in f]  # nearest
    t = not resnet
    out = _ffb(sd, "head.ref_3.", f[3], None, t, k, with_skip=False)
    out = _ffb(sd, "head.ref_2.", f[2], out, t, k)
    out = _ffb(sd, "head.ref_1.", f[1], out, t, k)
    out = _ffb(sd, "head.ref_0.", f[0], out, t, k)
    if not resnet:
        out = F.interpolate(out, scale_factor=4)
    out = F.conv2d(out, sd["head.out_conv.0.weight"], sd["head.out_conv.0.bias"], padding=1).relu()
    out = F.conv2d(out, sd["head.out_conv.2.weight"], sd["head.out_conv.2.bias"], padding=1)
    return F.interpolate(out, scale_factor=2)


# --------------------------------------------------------------------------- predictors
def depth_bin_prediction(logits: torch.Tensor, min_depth: float = 0.001, max_depth: float = 10, n_bins: int = 256) -> torch.Tensor:
    """probes.py:176-200 ('UD' bins, 'linear' norm): relu + 0.1, normalise over bins,
    expectation over ``linspace(min,max,n_bins)``."""
    bins = torch.linspace(min_depth, max_depth, n_bins)
    p = torch.relu(logits) + 0.1
    p = p / p.sum(dim=1, keepdim=True)
    return torch.einsum("ikhw,k->ihw", p, bins).unsqueeze(1)


def depth_sigmoid_prediction(x: torch.Tensor, min_depth: float = 0.001, max_depth: float = 10) -> torch.Tensor:
    """probes.py:209-212."""
    return min_depth + x.sigmoid() * (max_depth - min_depth)


def depth_head(sd, feats, head_type="linear", k=1, prediction_type="bindepth", min_depth=0.001, max_depth=10):
    """probes.py:153-157 DepthHead.forward."""
    x = linear_head(sd, feats, k) if head_type == "linear" else (multiscale_head(sd, list(feats)) if head_type == "multiscale" else dpt_head(sd, list(feats), k))
    if prediction_type == "bindepth":
        return depth_bin_prediction(x, min_depth, max_depth, 256)
    return depth_sigmoid_prediction(x, min_depth, max_depth)


def snorm_head(sd, feats, head_type="dpt", k=3):
    """probes.py:115-116 SurfaceNormalHead.forward."""
    if head_type == "multiscale":
        return multiscale_head(sd, list(feats))
    return linear_head(sd, feats, k) if head_type == "linear" else dpt_head(sd, list(feats), k)
