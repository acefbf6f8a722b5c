"""Oracle: one iteration of the train_depth.py / train_snorm.py hot loop, fp32 CPU.

Used (a) by the parity tests as the expected result of the HIP step and (b) by
``bench.py``'s ``cpu_baseline`` leg as the timed CPU "port" of the reference path.

Test infrastructure only (see oracle/__init__.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import losses, optim, probes, vit


class DepthProbeTrainer:
    """train_depth.py:93-144 for one process: frozen ViT under no_grad -> probe ->
    bilinear upsample to the target size -> DepthLoss -> backward -> AdamW -> LambdaLR."""

    def __init__(self, vit_sd, probe_sd, layers=(2, 5, 8, 11), heads=12, patch=16, head_type="linear", k=1,
                 prediction_type="bindepth", min_depth=0.001, max_depth=10, lr=5e-4, max_step=1000, warmup_step=150,
                 add_norm=True, scale_invariant=False):
        self.vit_sd = vit_sd
        self.probe_sd = {n: t.clone().requires_grad_(True) for n, t in probe_sd.items()}
        self.names = list(self.probe_sd)
        self.m = [torch.zeros_like(t) for t in self.probe_sd.values()]
        self.v = [torch.zeros_like(t) for t in self.probe_sd.values()]
        self.layers, self.heads, self.patch = tuple(layers), heads, patch
        self.head_type, self.k, self.prediction_type = head_type, k, prediction_type
        self.min_depth, self.max_depth = min_depth, max_depth
        self.base_lr, self.max_step, self.warmup_step = lr, max_step, warmup_step
        self.add_norm = add_norm
        self.scale_invariant = scale_invariant
        self.t = 0
        C = vit_sd["cls_token"].shape[-1]
        self.bn_running = [(torch.zeros(C), torch.ones(C)) for _ in layers]

    def lr_at(self, t: int) -> float:
        return self.base_lr * optim.cosine_decay_linear_warmup(t, self.max_step, self.warmup_step)

    def features(self, images: torch.Tensor):
        with torch.no_grad():
            f = vit.vit_dense_features(self.vit_sd, images, self.layers, self.heads, self.patch,
                                       add_norm=self.add_norm, bn_running=self.bn_running)
        return f if isinstance(f, list) else [f]

    def forward_loss(self, feats, target):
        pred = probes.depth_head(self.probe_sd, feats, self.head_type, self.k, self.prediction_type,
                                 self.min_depth, self.max_depth)
        pred = F.interpolate(pred, size=target.shape[-2:], mode="bilinear")
        if self.scale_invariant:  # train_depth.py:116-118
            from . import metrics

            pred = metrics.match_scale_and_shift(pred, target).clamp(min=0.001, max=1.0)
        return losses.depth_loss(pred, target), pred

    def step(self, images: torch.Tensor, target: torch.Tensor, grad_hook=None) -> float:
        feats = self.features(images)
        for p in self.probe_sd.values():
            p.grad = None
        loss, _ = self.forward_loss(feats, target)
        loss.backward()
        grads = [self.probe_sd[n].grad for n in self.names]
        if grad_hook is not None:
            grads = grad_hook(grads)
        lr = self.lr_at(self.t)  # LambdaLR: lr used by step t is lambda(t), t counted from 0
        self.t += 1
        with torch.no_grad():
            optim.adamw_step([self.probe_sd[n] for n in self.names], grads, self.m, self.v, self.t, lr)
        return float(loss.detach())


def synthetic_depth_batch(B: int, H: int, W: int, rank: int = 0, step: int = 0, zero_frac: float = 0.1):
    """SURVEY §8(d) synthetic inputs: images randn, depth U(0.05, 9.95) with a seeded
    ``zero_frac`` of pixels set to 0 so that the ``>0`` mask is exercised."""
    g = torch.Generator().manual_seed(1000 * rank + step)
    images = torch.randn(B, 3, H, W, generator=g)
    depth = torch.rand(B, 1, H, W, generator=g) * 9.9 + 0.05
    hole = torch.rand(B, 1, H, W, generator=g) < zero_frac
    depth[hole] = 0.0
    return images, depth


def synthetic_snorm_batch(B: int, H: int, W: int, rank: int = 0, step: int = 0, zero_frac: float = 0.1):
    images, depth = synthetic_depth_batch(B, H, W, rank, step, zero_frac)
    g = torch.Generator().manual_seed(7777 + 1000 * rank + step)
    n = torch.randn(B, 3, H, W, generator=g)
    n = n / n.norm(dim=1, keepdim=True).clamp_min(1e-6)
    return images, depth, n
