"""evals.utils.losses — drop-in for the losses the two trainers use
(evals/utils/losses.py:54-74,97-182), forward + analytic backward in HIP kernels."""
from __future__ import annotations

import torch.nn as nn

from mvp import functional as MF


class DepthLoss(nn.Module):
    """Reference: losses.py:97-111.  10*sig_loss + 0.5*gradient_loss, quirks Q1/Q2 included
    (target is modified in place)."""

    def __init__(self, weight_sig=10.0, weight_grad=0.5, max_depth=10):
        super().__init__()
        self.sig_w, self.grad_w, self.max_depth = weight_sig, weight_grad, max_depth

    def forward(self, pred, target):
        return MF.depth_loss(pred, target, self.sig_w, self.grad_w, self.max_depth)


def sig_loss(depth_pr, depth_gt, sigma=0.85, eps=0.001, only_mean=False):
    """Reference: losses.py:54-74 (as DepthLoss with the gradient term weighted 0).  ``only_mean`` is accepted and unused,
    exactly as in the reference (losses.py:54: the flag never reaches the arithmetic)."""
    if sigma != 0.85 or eps != 0.001:
        raise NotImplementedError("sig_loss: only the reference defaults are compiled in")
    return MF.depth_loss(depth_pr, depth_gt, 1.0, 0.0, float("inf"))


def angular_loss(snorm_pr, snorm_gt, mask, uncertainty_aware=False, eps=1e-4):
    """Reference: losses.py:157-182."""
    return MF.angular_loss(snorm_pr, snorm_gt, mask, uncertainty_aware, eps)
