"""The hot loop bodies of train_depth.py:93-144 and train_snorm.py:86-120, on the HIP path.

``train_depth_step`` / ``train_snorm_step`` are the per-batch bodies (what bench.py times);
``train`` mirrors the reference's ``train()`` signature for drop-in use.
"""
from __future__ import annotations

import torch

from . import functional as MF
from . import fused_step
from .pipeline import freeze_gc, pipelined_features


def extract_features(model, images, detach_model=True):
    """train_depth.py:103-112."""
    if detach_model:
        from .pipeline import GroupedFeatures

        def _detached(f):
            return [_f.detach() for _f in f] if isinstance(f, (tuple, list)) else f.detach()

        with torch.no_grad():
            feats = model(images)
            if isinstance(feats, GroupedFeatures):  # several batches stacked into one forward (mvp/pipeline.py): one result per batch
                feats = GroupedFeatures(_detached(f) for f in feats)
            else:
                feats = _detached(feats)
    else:
        # every head Function returns None for the feature gradient and the backbone engines run under no_grad: a
        # trainable backbone (optimizer.model_lr != 0, train_depth.py:566-575) would silently train nothing
        raise NotImplementedError("detach_model=False (backbone fine-tuning, optimizer.model_lr != 0) is outside the frozen-backbone hot path")
    return feats


def _finish_pending(optimizer):
    """FlatAdamW(overlap_comm=True): the previous step's gradient all-reduce has been running under this step's frozen
    forward; its AdamW update must land before the probe reads its weights."""
    fin = getattr(optimizer, "finish_pending", None)
    if fin is not None:
        fin()


def train_depth_step(model, probe, optimizer, scheduler, loss_fn, images, target, detach_model=True, scale_invariant=False, feats=None):
    """One iteration of train_depth.py:99-143; returns the loss as a DEVICE scalar (the
    reference's per-step ``loss.item()`` host sync is left to the caller).  ``feats``: the frozen features of ``images`` when
    the caller already has them in flight (mvp/pipeline.py); None runs the backbone here."""
    optimizer.zero_grad()
    if feats is None:
        feats = extract_features(model, images, detach_model)
    _finish_pending(optimizer)
    plan = fused_step.plan_for(probe, optimizer, scheduler, loss_fn, feats, target, scale_invariant)
    if plan is not None:  # the benchmarked probe: the same launches without the autograd tape (mvp/fused_step.py)
        return plan.run(feats, target, scheduler)
    pred = probe(feats)
    pred = MF.interpolate(pred, size=target.shape[-2:], mode="bilinear")
    if scale_invariant:  # train_depth.py:116-118: per-image scale/shift fit (detached) + clamp, one fused kernel each way
        from evals.utils.metrics import match_scale_and_shift

        pred = match_scale_and_shift(pred, target, clamp=(0.001, 1.0))
    loss = loss_fn(pred, target)
    MF.backward(loss)
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return loss.detach()


def train_snorm_step(model, probe, optimizer, scheduler, images, target, mask, detach_model=True, feats=None):
    """One iteration of train_snorm.py:93-120 (bicubic upsample, angular loss, UA iff 4 channels).  ``feats`` as in
    ``train_depth_step``."""
    from evals.utils.losses import angular_loss

    optimizer.zero_grad()
    if feats is None:
        feats = extract_features(model, images, detach_model)
    _finish_pending(optimizer)
    pred = probe(feats)
    pred = MF.interpolate(pred.contiguous(), size=target.shape[-2:], mode="bicubic")
    uncertainty = pred.shape[1] > 3
    loss = angular_loss(pred, target, mask, uncertainty_aware=uncertainty)
    MF.backward(loss)
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return loss.detach()


def _device_batches(loader, dev, keys=("image", "depth", "snorm")):
    """Wrap a host-side loader in the pinned double-buffered H2D prefetcher (mvp/prefetch.py); loaders that already
    yield device tensors (or custom iterables) pass through."""
    from torch.utils.data import DataLoader

    from .prefetch import DevicePrefetcher

    if isinstance(loader, DataLoader):
        return DevicePrefetcher(loader, dev, depth=2, keys=keys)
    return loader


def train(model, probe, train_loader, optimizer, scheduler, n_epochs, detach_model, loss_fn, rank=0, world_size=1,
          valid_loader=None, scale_invariant=False, wandb_use=False, is_final=False, is_navi=False, log_every=0):
    """Reference signature: train_depth.py:76-92.  Batches are dicts {"image", "depth"} (nyu.py:245-251)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    history = []
    freeze_gc()  # no full-heap collector pause inside the loop
    for ep in range(n_epochs):
        if world_size > 1 and hasattr(getattr(train_loader, "sampler", None), "set_epoch"):
            train_loader.sampler.set_epoch(ep)
        train_loss = 0.0
        if not detach_model:
            extract_features(model, None, detach_model)  # raises: backbone fine-tuning is outside the frozen hot path
        # the frozen forward of batch t+1 is already in flight (side stream) while the probe step of batch t runs
        for i, (batch, feats) in enumerate(pipelined_features(model, _device_batches(train_loader, dev), probe=probe)):
            target = batch["depth"].to(dev, non_blocking=True).contiguous()
            loss = train_depth_step(model, probe, optimizer, scheduler, loss_fn, None, target, detach_model, scale_invariant, feats=feats)
            train_loss += loss.item()  # the reference syncs every step too (train_depth.py:143)
        history.append(train_loss / max(len(train_loader), 1))
    # FlatAdamW(overlap_comm=True): the last step's all-reduce + AdamW are still pending — a caller that validates or saves
    # probe.state_dict() next must see the final weights (optimizer.state_dict() is not on that path)
    _finish_pending(optimizer)
    return history


def validate(model, probe, loader, loss_fn, verbose=True, scale_invariant=False, aggregate=True, render_images=False, wandb_use=False,
             is_navi=False, output_dir="result"):
    """Reference: validate(), train_depth.py:357-483 -> (mean loss, global_metrics, level_metrics).  Global metrics include the
    stuff/things groups when ``is_navi`` is False (batches then carry "segmentation", nyu.py:245-251); per-segment rows are
    collected in ``validate.last_segment_metrics`` (the reference only plots them).  PNG dumps / W&B are out of scope."""
    from evals.utils.metrics import evaluate_depth

    dev = torch.device("cuda", torch.cuda.current_device())
    total_loss, count = 0.0, 0
    global_metrics, level_metrics, segments = None, None, []
    with torch.no_grad():
        # eval-mode forwards mutate nothing: the next batches' forwards run on side streams under this batch's probe + metric kernels
        for batch, feats in pipelined_features(model, _device_batches(loader, dev, keys=("image", "depth", "snorm", "segmentation")), probe=probe):
            target = batch["depth"].to(dev).contiguous()
            seg = None if is_navi else batch["segmentation"].to(dev)
            pred = probe(feats)
            pred = MF.interpolate(pred, size=target.shape[-2:], mode="bilinear")
            loss = loss_fn(pred, target)
            total_loss += loss.item()
            gm, lm, sm = evaluate_depth(pred, target, seg, scale_invariant=scale_invariant, nyu_crop=not is_navi, is_navi=is_navi)
            segments.extend(sm)
            if global_metrics is None:
                global_metrics = {k: [v.reshape(-1)] for k, v in gm.items()}
                level_metrics = {L: {k: [v] for k, v in d.items()} for L, d in lm.items()}
            else:
                for k, v in gm.items():
                    global_metrics[k].append(v.reshape(-1))
                for L, d in lm.items():
                    for k, v in d.items():
                        level_metrics[L][k].append(v)
            count += 1
    if global_metrics is None:
        raise ValueError("validate(): empty loader")
    if aggregate:
        global_metrics = {k: torch.cat(v, dim=0).mean() for k, v in global_metrics.items()}
        level_metrics = {L: {k: torch.cat(v, dim=0).mean() for k, v in d.items()} for L, d in level_metrics.items()}
    validate.last_segment_metrics = segments
    return total_loss / max(count, 1), global_metrics, level_metrics
