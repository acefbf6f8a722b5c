"""evals.utils.metrics — drop-in for the GLOBAL validation metrics of the reference
(evals/utils/metrics.py:106-178, 397-440, 742-780) as fused HIP reductions.

The stuff/things, centroid-level and per-segment breakdowns of the reference need OneFormer
panoptic maps (data_processing/Oneformer_preprocess) and are out of scope: evaluate_* return empty
dicts / lists in their place (same tuple arity as the reference)."""
from __future__ import annotations

import torch

from mvp import lib

_DEPTH_KEYS = ["d1", "d2", "d3", "rmse", "mean_pred", "std_pred", "variance_pred", "mean_gt", "std_gt", "variance_gt", "variance_ratio"]


def _ws(B, dev):
    return torch.empty(int(lib.load().mvp_metrics_workspace_bytes(B)) // 4 + 4, dtype=torch.float32, device=dev)


def _depth_metrics(depth_pr, depth_gt, scale_invariant):
    assert depth_pr.shape == depth_gt.shape, f"{depth_pr.shape} != {depth_gt.shape}"
    if not depth_pr.is_cuda:
        raise lib.MvpError("metrics need device tensors (no CPU fallback)")
    B = depth_pr.shape[0]
    p = depth_pr.detach().reshape(B, -1).contiguous().float()
    g = depth_gt.detach().reshape(B, -1).contiguous().float()
    out = torch.empty(B, 12, dtype=torch.float32, device=p.device)
    ss = torch.empty(B, 2, dtype=torch.float32, device=p.device)
    ws = _ws(B, p.device)
    a = lib.DepthMetricsArgs(lib.ptr(p), lib.ptr(g), lib.ptr(out), lib.ptr(ss), lib.ptr(ws), ws.numel() * 4, B, p.shape[1], int(scale_invariant))
    lib.call("mvp_depth_metrics", a)
    return out, ss


def match_scale_and_shift(prediction, target):
    """Reference: metrics.py:742-780 (per-image least-squares scale & shift over target > 0)."""
    assert len(target.shape) == len(prediction.shape)
    _, ss = _depth_metrics(prediction, target, True)
    shape = (-1,) + (1,) * (prediction.ndim - 1)
    return prediction * ss[:, 0].view(shape) + ss[:, 1].view(shape)


def evaluate_depth(depth_pr, depth_gt, segmentation_map=None, image_average=False, scale_invariant=False, nyu_crop=False, num_levels=5, is_navi=False):
    """Reference: metrics.py:106-358.  Returns (global_metrics, metrics_by_level, segment_metrics);
    the last two are empty (see module docstring)."""
    out, _ = _depth_metrics(depth_pr, depth_gt, scale_invariant)
    out = out.cpu()
    gm = {k: out[:, i] for i, k in enumerate(_DEPTH_KEYS)}
    for k in ("mean_pred", "mean_gt"):  # the reference returns these as [B,1,1]
        gm[k] = gm[k].view(-1, 1, 1)
    if image_average:
        gm = {k: v.mean() for k, v in gm.items()}
    return gm, {}, []


def evaluate_surface_norm(snorm_pr, snorm_gt, segmentation_map=None, image_average=False, num_levels=5, thresh=[11.25, 22.5, 30.0], is_navi=False):
    """Reference: metrics.py:397-577 (global part)."""
    if not snorm_pr.is_cuda:
        raise lib.MvpError("metrics need device tensors (no CPU fallback)")
    B, Cp = snorm_pr.shape[:2]
    assert snorm_pr[:, :3].shape == snorm_gt.shape, f"{snorm_pr[:, :3].shape} != {snorm_gt.shape}"
    p = snorm_pr.detach().reshape(B, Cp, -1).contiguous().float()
    g = snorm_gt.detach().reshape(B, 3, -1).contiguous().float()
    out = torch.empty(B, 5, dtype=torch.float32, device=p.device)
    ws = _ws(B, p.device)
    a = lib.SnormMetricsArgs(lib.ptr(p), lib.ptr(g), lib.ptr(out), lib.ptr(ws), ws.numel() * 4, B, Cp, p.shape[2], thresh[0], thresh[1], thresh[2])
    lib.call("mvp_snorm_metrics", a)
    out = out.cpu()
    gm = {k: out[:, i] for i, k in enumerate(["d1", "d2", "d3", "rmse"])}
    if image_average:
        gm = {k: v.mean() for k, v in gm.items()}
    return gm, {}, []
