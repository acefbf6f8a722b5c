"""Oracle: LR schedule and AdamW step of train_depth.py:624-641, CPU.

Test infrastructure only (see oracle/__init__.py).
"""
from __future__ import annotations

import math
from typing import List

import torch


def cosine_decay_linear_warmup(step: float, max_step: float, warmup_step: float, min_factor: float = 0.01) -> float:
    """evals/utils/optim.py:124-133."""
    assert max_step > warmup_step
    span = 1 - min_factor
    if step <= warmup_step:
        return span * (step / warmup_step) + min_factor
    rel = (step - warmup_step) / (max_step - warmup_step)
    return span * math.cos(0.5 * rel * math.pi) + min_factor


def adamw_step(params: List[torch.Tensor], grads: List[torch.Tensor], m: List[torch.Tensor], v: List[torch.Tensor],
               step: int, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, wd: float = 0.01) -> None:
    """torch.optim.AdamW defaults as used by train_depth.py:624-627 (decoupled decay,
    bias-corrected, eps added after sqrt(v_hat)).  ``step`` is 1-based."""
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    for p, g, mi, vi in zip(params, grads, m, v):
        p.mul_(1 - lr * wd)
        mi.mul_(beta1).add_(g, alpha=1 - beta1)
        vi.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (vi.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(mi, denom, value=-lr / bc1)
