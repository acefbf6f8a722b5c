"""Oracle: global validation metrics of evals/utils/metrics.py, fp32 CPU.  Test infrastructure only.

evaluate_depth (global part, metrics.py:106-178), match_scale_and_shift (metrics.py:742-780),
evaluate_surface_norm (global part, metrics.py:397-440), and the stuff/things, centroid-level and per-segment
breakdowns of both (metrics.py:179-358, 441-577) given an integer segmentation map.  Pinned by tests/golden/metrics.npz and
metrics_seg.npz (outputs of the reference's own evaluate_depth / evaluate_surface_norm, tests/golden/make_goldens.py)."""
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
    out = prediction * scale.view(-1, 1, 1).detach() + shift.view(-1, 1, 1).detach()  # metrics.py:775-776: no gradient through the fit
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


# OneFormer ADE20K-150 panoptic ids (evals/utils/oneformer_id2label.py:154-303): 22 "stuff" classes; every other id is a
# "thing" except 11, 17, 40 and 68, which the reference lists in neither group.
STUFF = [0, 1, 2, 3, 4, 5, 6, 9, 13, 16, 21, 26, 29, 46, 52, 60, 91, 94, 96, 106, 113, 128]
THINGS = [i for i in range(150) if i not in STUFF and i not in (11, 17, 40, 68)]


def level_masks(valid, num_levels=5):
    """metrics.py:249-270 / 446-460: nested centre boxes ("centroid levels"), each minus the earlier ones, times valid.
    The offset is derived from the HEIGHT and applied to both axes."""
    H, W = valid.shape[-2:]
    cum = torch.zeros_like(valid)
    out = []
    for level in range(1, num_levels + 1):
        m = torch.zeros_like(valid)
        off = (H // num_levels) * (num_levels - level) // 2
        m[..., off:H - off, off:W - off] = 1
        m = (m - cum).clamp(min=0) * valid
        cum = cum + m
        out.append(m)
    return out


def _safe(n, mode):
    return torch.where(n == 0, torch.tensor(1e-6), n) if mode == "eps" else n.clamp(min=1)


def depth_breakdown(pred, gt, seg, scale_invariant=False, num_levels=5):
    """evaluate_depth beyond the global block (metrics.py:179-358): stuff/things metrics, metrics_by_level, segment_metrics."""
    if pred.ndim == 4:
        pred, gt = pred.squeeze(1), gt.squeeze(1)
    if scale_invariant:
        pred = match_scale_and_shift(pred, gt)
    valid = (gt > 0).float()
    pred = pred * valid
    thresh = torch.maximum(gt / pred.clamp(min=1e-9), pred / gt.clamp(min=1e-9))
    sse = (gt - pred) ** 2
    hits = [(thresh < 1.25 ** k).float() for k in (1, 2, 3)]
    groups = {}
    for name, ids in (("stuff", STUFF), ("things", THINGS)):
        m = torch.isin(seg, torch.tensor(ids)).float() * valid
        n = _safe(m.sum((1, 2)), "eps")
        for k in range(3):
            groups[f"{name}_d{k + 1}"] = (hits[k] * m).sum((1, 2)) / n
        groups[f"{name}_rmse"] = ((sse * m).sum((1, 2)) / n).sqrt()
        groups[f"{name}_pixels"] = n
    levels = {}
    for i, m in enumerate(level_masks(valid, num_levels)):
        n = _safe(m.sum((1, 2)), "eps")
        levels[f"level_{i + 1}"] = {**{f"d{k + 1}": (hits[k] * m).sum((1, 2)) / n for k in range(3)}, "rmse": ((sse * m).sum((1, 2)) / n).sqrt()}
    segments = []
    for sid in torch.unique(seg):
        m = (seg == sid).float() * valid
        area = _safe(m.sum((1, 2)), "eps")
        d1 = (hits[0] * m).sum((1, 2)) / area
        segments += [(int(sid), b, float(area[b]), float(d1[b])) for b in range(pred.shape[0])]
    return groups, levels, segments


def snorm_breakdown(pred, gt, seg, thresh=(11.25, 22.5, 30.0), num_levels=5):
    """evaluate_surface_norm beyond the global block (metrics.py:441-560).  Quirk kept: stuff/things "rmse" is
    sqrt(sum err^2) / pixels (metrics.py:499,513), not sqrt(sum / pixels)."""
    pred = pred[:, :3]
    err = torch.acos(torch.cosine_similarity(pred, gt, dim=1).clamp(-1, 1)) * 180.0 / torch.pi
    valid = (gt.abs().sum(1) > 0).float()
    err = err * valid
    hits = [(err < t).float() for t in thresh]
    levels = {}
    for i, m in enumerate(level_masks(valid, num_levels)):
        n = _safe(m.sum((1, 2)), "one")
        levels[f"level_{i + 1}"] = {**{f"d{k + 1}": (hits[k] * m).sum((1, 2)) / n for k in range(3)},
                                    "rmse": (((err * m) ** 2).sum((1, 2)) / n).sqrt()}
    groups = {}
    for name, ids in (("stuff", STUFF), ("things", THINGS)):
        m = torch.isin(seg, torch.tensor(ids)).float() * valid
        n = _safe(m.sum((1, 2)), "one")
        for k in range(3):
            groups[f"{name}_d{k + 1}"] = (hits[k] * m).sum((1, 2)) / n
        groups[f"{name}_rmse"] = (err ** 2 * m).sum((1, 2)).sqrt() / n
        groups[f"{name}_pixels"] = n
    segments = []
    for sid in torch.unique(seg):
        m = (seg == sid).float() * valid
        area = _safe(m.sum((1, 2)), "one")
        d1 = (hits[0] * m).sum((1, 2)) / area
        segments += [(int(sid), b, float(area[b]), float(d1[b])) for b in range(pred.shape[0])]
    return groups, levels, segments
