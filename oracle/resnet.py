"""Oracle: ResNet-50 (torchvision v1.5 layout) dense multi-layer feature extraction, fp32 CPU.

Restates DINO_RESNET.forward / MoCoV3_RES.forward (evals/models/dino_res50.py:83-101,
mocov3_res50.py:97-116): Resize((fixed,fixed)) -> [conv1,bn1,relu,maxpool], layer1..4 with
eval-mode internal BatchNorms, train-mode BatchNorm2d taps.

PARITY UNPINNED for the trunk arithmetic: torchvision 0.17.1 (README.md:59) is not vendored in
/root/reference and not installed here; the Bottleneck below follows the published v1.5
architecture (stride on the 3x3 conv), keys follow torchvision's state-dict layout
(conv1, bn1, layer{1-4}.{i}.{conv,bn}{1,2,3}, downsample.{0,1}).

Test infrastructure only (see oracle/__init__.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

StateDict = Dict[str, torch.Tensor]
LAYERS = (3, 4, 6, 3)
WIDTHS = (64, 128, 256, 512)


def make_resnet50_weights(seed: int = 0, width_div: int = 1) -> StateDict:
    """Seeded weights in torchvision's key layout (kaiming-normal fan_out convs as torchvision
    initialises them; BN affine / running stats randomised so that BN folding is exercised).
    ``width_div`` shrinks all widths (tests)."""
    g = torch.Generator().manual_seed(seed)
    sd: StateDict = {}

    def conv(name, cout, cin, k):
        std = math.sqrt(2.0 / (cout * k * k))
        sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * std

    def bn(name, c):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_var"] = 1.0 + 0.2 * torch.rand(c, generator=g)

    w0 = 64 // width_div
    conv("conv1", w0, 3, 7)
    bn("bn1", w0)
    inp = w0
    for li, (n, wd) in enumerate(zip(LAYERS, WIDTHS), start=1):
        wd = wd // width_div
        for bi in range(n):
            p = f"layer{li}.{bi}."
            conv(p + "conv1", wd, inp, 1); bn(p + "bn1", wd)
            conv(p + "conv2", wd, wd, 3); bn(p + "bn2", wd)
            conv(p + "conv3", wd * 4, wd, 1); bn(p + "bn3", wd * 4)
            if bi == 0:
                conv(p + "downsample.0", wd * 4, inp, 1); bn(p + "downsample.1", wd * 4)
            inp = wd * 4
    return sd


def _bn_eval(sd, name, x, eps=1e-5):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"], False, 0.0, eps)


def bottleneck(sd: StateDict, p: str, x: torch.Tensor, stride: int) -> torch.Tensor:
    """torchvision Bottleneck v1.5: 1x1 -> 3x3 (stride) -> 1x1 (x4), BN after each conv,
    ReLU after the first two and after the residual add."""
    idt = x
    out = _bn_eval(sd, p + "bn1", F.conv2d(x, sd[p + "conv1.weight"])).relu()
    out = _bn_eval(sd, p + "bn2", F.conv2d(out, sd[p + "conv2.weight"], stride=stride, padding=1)).relu()
    out = _bn_eval(sd, p + "bn3", F.conv2d(out, sd[p + "conv3.weight"]))
    if p + "downsample.0.weight" in sd:
        idt = _bn_eval(sd, p + "downsample.1", F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride))
    return (out + idt).relu()


def stage(sd: StateDict, i: int, x: torch.Tensor) -> torch.Tensor:
    """self.layers[i] of dino_res50.py:38-51."""
    if i == 0:
        x = _bn_eval(sd, "bn1", F.conv2d(x, sd["conv1.weight"], stride=2, padding=3)).relu()
        return F.max_pool2d(x, 3, 2, 1)
    n = LAYERS[i - 1]
    for bi in range(n):
        x = bottleneck(sd, f"layer{i}.{bi}.", x, stride=(2 if (bi == 0 and i > 1) else 1))
    return x


def resize_fixed(x: torch.Tensor, size: int) -> torch.Tensor:
    """torchvision.transforms.Resize((size,size)) on a tensor = interpolate(bilinear,
    align_corners=False, antialias=True) (dino_res50.py:80,85)."""
    if x.shape[-2:] == (size, size):
        return x
    return F.interpolate(x, size=(size, size), mode="bilinear", align_corners=False, antialias=True)


def resnet_dense_features(sd: StateDict, images: torch.Tensor, multilayers: Sequence[int], fixed_size: int = 480, add_norm: bool = True,
                          bn_affine=None, bn_running=None, bn_training: bool = True, eps: float = 1e-5):
    """DINO_RESNET.forward (dino_res50.py:83-101).  ``bn_affine[i]`` / ``bn_running[i]`` are
    indexed by LAYER index i (the wrapper holds 5 BatchNorm2d modules, one per stage)."""
    x = resize_fixed(images, fixed_size)
    outs = []
    for i in range(5):
        x = stage(sd, i, x)
        if i in multilayers:
            if add_norm:
                w, b = bn_affine[i] if bn_affine is not None else (None, None)
                rm, rv = bn_running[i] if bn_running is not None else (None, None)
                if bn_training:
                    if rm is None:
                        rm, rv = torch.zeros(x.shape[1]), torch.ones(x.shape[1])
                    outs.append(F.batch_norm(x, rm, rv, w, b, True, 0.1, eps))
                else:
                    outs.append(F.batch_norm(x, rm, rv, w, b, False, 0.1, eps))
            else:
                outs.append(x)
        if i == max(multilayers):
            break
    return outs[0] if len(outs) == 1 else outs
