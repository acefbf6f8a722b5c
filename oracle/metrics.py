"""Oracle: global validation metrics of evals/utils/metrics.py, fp32 CPU.  Test infrastructure only.

evaluate_depth (global part, metrics.py:106-178), match_scale_and_shift (metrics.py:742-780),
evaluate_surface_norm (global part, metrics.py:397-440).  The stuff/things, centroid-level and
per-segment breakdowns need OneFormer panoptic maps (data_processing/) and are out of scope."""
from __future__ import annotations

import torch


def match_scale_and_shift(prediction, target):
    four = prediction.ndim == 4
    if four:
        prediction, target = prediction.squeeze(1), target.squeeze(1)
    mask = (target > 0).float()
    a00 = (mask * prediction * prediction).sum((1, 2))
    a01 = (mask * prediction).sum((1, 2))
    a11 = mask.sum((1, 2))
    b0 = (mask * prediction * target).sum((1, 2))
    b1 = (mask * target).sum((1, 2))
    det = a00 * a11 - a01 * a01
    ok = det != 0
    scale, shift = torch.ones_like(b0), torch.zeros_like(b1)
    scale[ok] = (a11[ok] * b0[ok] - a01[ok] * b1[ok]) / det[ok]
    shift[ok] = (-a01[ok] * b0[ok] + a00[ok] * b1[ok]) / det[ok]
    out = prediction * scale.view(-1, 1, 1) + shift.view(-1, 1, 1)
    return out[:, None] if four else out


def depth_global_metrics(pred, gt, scale_invariant=False):
    if pred.ndim == 4:
        pred, gt = pred.squeeze(1), gt.squeeze(1)
    if scale_invariant:
        pred = match_scale_and_shift(pred, gt)
    valid = (gt > 0).float()
    pred = pred * valid
    n = valid.sum((1, 2))
    n = torch.where(n == 0, torch.tensor(1e-6), n)
    mean_p = (pred * valid).sum((1, 2)) / n
    var_p = (((pred - mean_p.view(-1, 1, 1)) ** 2) * valid).sum((1, 2)) / n
    mean_g = (gt * valid).sum((1, 2)) / n
    var_g = (((gt - mean_g.view(-1, 1, 1)) ** 2) * valid).sum((1, 2)) / n
    thresh = torch.maximum(gt / pred.clamp(min=1e-9), pred / gt.clamp(min=1e-9))
    out = {f"d{k}": ((thresh < 1.25 ** k).float() * valid).sum((1, 2)) / n for k in (1, 2, 3)}
    out["rmse"] = (((gt - pred) ** 2 * valid).sum((1, 2)) / n).sqrt()
    out.update(mean_pred=mean_p, std_pred=var_p.sqrt(), variance_pred=var_p, mean_gt=mean_g, std_gt=var_g.sqrt(), variance_gt=var_g,
               variance_ratio=var_p / torch.where(var_g == 0, torch.tensor(1e-6), var_g))
    return out


def snorm_global_metrics(pred, gt, thresh=(11.25, 22.5, 30.0)):
    pred = pred[:, :3]
    cos = torch.cosine_similarity(pred, gt, dim=1).clamp(-1, 1)
    err = torch.acos(cos) * 180.0 / torch.pi
    valid = (gt.abs().sum(1) > 0).float()
    err = err * valid
    n = valid.sum((1, 2)).clamp(min=1)
    out = {f"d{i + 1}": ((err < t).float() * valid).sum((1, 2)) / n for i, t in enumerate(thresh)}
    out["rmse"] = (err.pow(2).sum((1, 2)) / n).sqrt()
    return out
