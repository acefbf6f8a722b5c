"""Oracle: training losses of train_depth.py / train_snorm.py, fp32 CPU.

Test infrastructure only (see oracle/__init__.py).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def sig_loss(pred: torch.Tensor, target: torch.Tensor, sigma: float = 0.85, eps: float = 1e-3) -> torch.Tensor:
    """losses.py:54-74 — g = log(p+eps) - log(t+eps) over valid (t>0) pixels;
    sqrt(mean(g^2) - sigma * mean(g)^2)."""
    valid = target > 0
    g = torch.log(pred[valid] + eps) - torch.log(target[valid] + eps)
    return (g.pow(2).mean() - sigma * g.mean().pow(2)).sqrt()


def gradient_loss(pred: torch.Tensor, target: torch.Tensor, eps: float = 1e-3) -> torch.Tensor:
    """losses.py:114-154, quirk Q1 reproduced: the strided "downscales" and the +-2
    differences index dims 0/1 of the [B,1,H,W] tensors (batch / channel), not H/W.
    Written out explicitly: for batch strides s in {1,2,4,6} take the sub-batch
    b = 0, s, 2s, ...; the "vertical" term pairs sub-batch entries j and j+2; the
    "horizontal" term slices the size-1 channel dim and is always empty."""
    total = pred.new_zeros(())
    for s in (1, 2, 4, 6):
        p = pred[::s, ::s] if s > 1 else pred
        t = target[::s, ::s] if s > 1 else target
        valid = t > 0
        n = valid.sum()
        d = (torch.log(p + eps) - torch.log(t + eps)) * valid
        v = (d[:-2] - d[2:]).abs() * (valid[:-2] * valid[2:])
        h = (d[:, :-2] - d[:, 2:]).abs() * (valid[:, :-2] * valid[:, 2:])
        total = total + (h.sum() + v.sum()) / n
    return total


def depth_loss(pred: torch.Tensor, target: torch.Tensor, w_sig: float = 10.0, w_grad: float = 0.5, max_depth: float = 10) -> torch.Tensor:
    """losses.py:97-111 DepthLoss.forward.  Mutates ``target`` in place (quirk Q2)."""
    target[target > max_depth] = 0
    return w_sig * sig_loss(pred, target) + w_grad * gradient_loss(pred, target)


def angular_loss(pred: torch.Tensor, gt: torch.Tensor, mask: torch.Tensor, uncertainty_aware: bool = False, eps: float = 1e-4) -> torch.Tensor:
    """losses.py:157-182."""
    assert mask.ndim == 4
    m = mask.squeeze(1).float()
    if uncertainty_aware:
        assert pred.shape[1] == 4
        ang = torch.cosine_similarity(pred[:, :3], gt, dim=1).clamp(-1 + eps, 1 - eps).acos()
        kappa = F.elu(pred[:, 3]) + 1.01
        reg = (1 + (-kappa * math.pi).exp()).log() - (kappa.pow(2) + 1).log()
        loss = reg + kappa * ang
    else:
        assert pred.shape[1] == 3
        loss = torch.cosine_similarity(pred, gt, dim=1).clamp(-1 + eps, 1 - eps).acos()
    return loss[m.bool()].mean()
