"""The probe step of train_depth.py:113-142 without the autograd tape, for the benchmarked probe: DepthHead(linear, k = 1, bindepth)
+ bilinear upsample + DepthLoss + FlatAdamW + LambdaLR.

Why: beside a frozen forward the step is device-bound, but alone — the steps that follow the last forward of an epoch, and every
step once the forward gets faster — it is HOST-bound: ~30 launches cost 0.26-0.30 ms of Python (three ``autograd.Function.apply``
and the engine's walk back through them, ``nn.Module.__call__`` twice, torch's wrappers around ``optimizer.step`` and
``scheduler.step``: profiles/r04_host_profile.txt) for less device time than that.  The chain of this probe is static — forward
GEMM, bins + x4 upsample, upsample to the target, loss with its analytic gradient, the three adjoints, AdamW — so the tape buys
nothing: ``LinearBinsDepthStep.run`` issues the SAME launches with the SAME arguments in the same order (it calls the functions the
autograd Functions are made of: ``functional.linear_bins_forward`` / ``linear_bins_backward``, ``ops.resize``, ``ops.depth_loss``),
writes the gradients into FlatAdamW's flat buffer and steps.  Bit-identical trajectories (tests/test_gpu_fused_step.py); every
other probe / loss / optimiser, ``scale_invariant``, registered hooks, or ``MVP_FUSED_STEP=0`` take the tape as before.
"""
from __future__ import annotations

import os
import weakref

import torch

from . import functional as MF
from . import lib, ops


def enabled() -> bool:
    return os.environ.get("MVP_FUSED_STEP", "1") != "0"


def _no_hooks(*mods) -> bool:
    for m in mods:
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
            return False
    return True


def fast_scheduler_step(scheduler, optimizer) -> None:
    """``scheduler.step()``.  For a plain one-group ``LambdaLR`` (train_depth.py:636-641) past its first call, the same state
    transition without the wrappers (``_step_count``, ``last_epoch``, ``group["lr"]`` = base_lr * lambda(last_epoch), ``_last_lr``:
    torch/optim/lr_scheduler.py ``LRScheduler.step`` / ``_update_lr`` / ``LambdaLR.get_lr`` — the same float expression, so the
    same lr bit for bit; tests/test_fused_step_cpu.py holds the two side by side, ``state_dict()`` included)."""
    if (type(scheduler) is torch.optim.lr_scheduler.LambdaLR and scheduler._step_count > 1 and len(optimizer.param_groups) == 1
            and len(scheduler.lr_lambdas) == 1 and not isinstance(optimizer.param_groups[0]["lr"], torch.Tensor)):
        scheduler._step_count += 1
        scheduler.last_epoch += 1
        lr = scheduler.base_lrs[0] * scheduler.lr_lambdas[0](scheduler.last_epoch)
        optimizer.param_groups[0]["lr"] = lr
        scheduler._last_lr = [lr]
        return
    scheduler.step()


class LinearBinsDepthStep:
    """One (probe, optimizer, loss) triple's tape-free step.  ``plan_for`` hands out the cached instance or None."""

    def __init__(self, probe, optimizer, loss_fn):
        conv = probe.head.conv
        self.probe, self.opt, self.loss_fn = weakref.ref(probe), weakref.ref(optimizer), weakref.ref(loss_fn)
        self.weight, self.bias = conv.weight, conv.bias
        slots = dict((id(p), v) for p, v in optimizer.grad_slots())
        self.gw, self.gb = slots[id(self.weight)], slots[id(self.bias)]
        self.K = int(self.weight.shape[0])

    def still_valid(self, probe, optimizer, loss_fn) -> bool:
        if self.probe() is not probe or self.opt() is not optimizer or self.loss_fn() is not loss_fn:
            return False
        conv = probe.head.conv
        return (conv.weight is self.weight and conv.bias is self.bias and self.weight.requires_grad and self.bias.requires_grad
                and _no_hooks(probe, probe.head, conv, probe.predict, loss_fn)
                and not optimizer._optimizer_step_pre_hooks and not optimizer._optimizer_step_post_hooks
                and not self.weight._backward_hooks and not self.bias._backward_hooks)

    def run(self, feats, target, scheduler):
        """-> the loss as a device scalar.  The caller has run ``optimizer.zero_grad()`` and ``finish_pending()`` (mvp/train.py)."""
        probe, opt, loss_fn = self.probe(), self.opt(), self.loss_fn()
        pr = probe.head.precision
        mn, mx = float(probe.predict.min_depth), float(probe.predict.max_depth)
        K = self.K
        with torch.no_grad():
            pack = MF.pack_features(feats if type(feats) is list else list(feats), pr)
            # ---- forward: probes.py:153-157 (head + bins), train_depth.py:114 (bilinear to the target's size), losses.py:97-111
            depth, inv, gate = MF.linear_bins_forward(self.weight, self.bias, pack, pr, mn, mx)
            B, _, Hi, Wi = depth.shape
            Ho, Wo = int(target.shape[-2]), int(target.shape[-1])
            dev = depth.device
            pred = torch.empty(B, 1, Ho, Wo, dtype=torch.float32, device=dev)
            ops.resize(depth, pred, B, Hi, Wi, Ho, Wo, lib.RESIZE_BILINEAR, align_corners=False, scale_h=0.0, scale_w=0.0)
            HW = Ho * Wo
            out = torch.empty(4, dtype=torch.float32, device=dev)
            grad = torch.empty_like(pred)
            ws = torch.empty(ops.depth_loss_workspace_bytes(B, HW) // 4 + 4, dtype=torch.float32, device=dev)
            ops.depth_loss(pred, target, out, grad, ws, B, HW, float(loss_fn.sig_w), float(loss_fn.grad_w), float(loss_fn.max_depth))
            # ---- backward: the adjoints in the tape's order (loss gradient as stored, resize, bins + x4, weight / bias gradients)
            gx = torch.empty(B, 1, Hi, Wi, dtype=torch.float32, device=dev)
            ops.resize(grad, gx, B, Hi, Wi, Ho, Wo, lib.RESIZE_BILINEAR, align_corners=False, scale_h=0.0, scale_w=0.0, backward=True)
            dW, db = MF.linear_bins_backward(gx, depth, inv, gate, pack, pr, K, mn, mx, self.gw.view(K, -1), self.gb)
            if dW.data_ptr() != self.gw.data_ptr():  # a channel-padded packing: the gradient came back in a buffer of its own
                self.gw.view(K, -1).copy_(dW)
            self.weight.grad, self.bias.grad = self.gw, self.gb  # what autograd leaves behind (FlatAdamW.zero_grad dropped them)
            opt.step_in_place()
            if scheduler is not None:
                fast_scheduler_step(scheduler, opt)
        return out[0]


def plan_for(probe, optimizer, scheduler, loss_fn, feats, target, scale_invariant: bool):
    """The tape-free step for this call of ``train_depth_step``, or None (the generic tape path)."""
    if scale_invariant or not enabled() or not torch.is_grad_enabled():
        return None
    plan = getattr(optimizer, "_mvp_fused_plan", None)
    if plan is False:  # this optimiser's triple was looked at and does not qualify
        return None
    if plan is None or not plan.still_valid(probe, optimizer, loss_fn):
        plan = _build(probe, optimizer, loss_fn)
        optimizer._mvp_fused_plan = plan if plan is not None else False
        if plan is None:
            return None
    if not (isinstance(target, torch.Tensor) and target.is_cuda and target.is_contiguous() and target.dtype == torch.float32):
        return None  # the tape path raises the descriptive error
    if isinstance(feats, torch.Tensor):
        return None
    return plan


def _build(probe, optimizer, loss_fn):
    from evals.models.probes import DepthBinPrediction, DepthHead, Linear
    from evals.utils.losses import DepthLoss

    from .optim import FlatAdamW

    if type(probe) is not DepthHead or type(loss_fn) is not DepthLoss or not isinstance(optimizer, FlatAdamW):
        return None
    head = probe.head
    if not (isinstance(head, Linear) and head.kernel_size == 1 and isinstance(probe.predict, DepthBinPrediction)
            and head.conv.out_channels % 8 == 0 and head.conv.bias is not None):
        return None
    w, b = head.conv.weight, head.conv.bias
    params = optimizer._params
    if len(params) != 2 or not any(p is w for p in params) or not any(p is b for p in params) or w.dtype != torch.float32:
        return None
    plan = LinearBinsDepthStep(probe, optimizer, loss_fn)
    return plan if plan.still_valid(probe, optimizer, loss_fn) else None
