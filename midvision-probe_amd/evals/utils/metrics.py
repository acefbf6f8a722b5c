"""evals.utils.metrics — drop-in for the validation metrics of the reference (evals/utils/metrics.py:106-358, 397-577,
742-780) as fused HIP reductions: the global metrics (one masked multi-metric reduction per image), and the centroid-level,
stuff/things and per-segment breakdowns (ONE segmented masked reduction per batch, csrc/metrics.hip mb_partial, instead of the
reference's per-level / per-segment full-image passes).  Return values have the reference's structure:
(global_metrics: dict of [B] CPU tensors, metrics_by_level: {"level_k": {d1,d2,d3,rmse}}, segment_metrics: list of dicts)."""
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


class _ScaleShift(torch.autograd.Function):
    """y = x * scale[b] + shift[b] (optionally clamped); scale / shift are constants of the tape, as in the reference."""

    @staticmethod
    def forward(ctx, x, ss, lo, hi, clamp):
        B = x.shape[0]
        xc = x.detach().contiguous().float()
        y = torch.empty_like(xc)
        lib.call("mvp_scale_shift", lib.ScaleShiftArgs(lib.ptr(xc), lib.ptr(ss), None, lib.ptr(y), B, xc.numel() // B, lo, hi, int(clamp), 0))
        ctx.save_for_backward(xc, ss)
        ctx.cfg = (lo, hi, clamp)
        return y

    @staticmethod
    def backward(ctx, gy):
        xc, ss = ctx.saved_tensors
        lo, hi, clamp = ctx.cfg
        B = xc.shape[0]
        g = gy.contiguous().float()
        gx = torch.empty_like(xc)
        lib.call("mvp_scale_shift", lib.ScaleShiftArgs(lib.ptr(xc), lib.ptr(ss), lib.ptr(g), lib.ptr(gx), B, xc.numel() // B, lo, hi, int(clamp), 1))
        return gx, None, None, None, None


def match_scale_and_shift(prediction, target, clamp=None):
    """Reference: metrics.py:742-780 (per-image least-squares scale & shift over target > 0; the fitted scale / shift are detached,
    so gradients reach ``prediction`` only through the final affine map).  ``clamp=(lo, hi)`` fuses the ``.clamp`` that follows it in
    the scale-invariant training branch (train_depth.py:116-118)."""
    assert len(target.shape) == len(prediction.shape)
    _, ss = _depth_metrics(prediction, target, True)
    lo, hi = (float(clamp[0]), float(clamp[1])) if clamp is not None else (0.0, 0.0)
    return _ScaleShift.apply(prediction, ss, lo, hi, clamp is not None)


def _breakdown(pred, gt, seg, scale_shift, Cp, num_levels, thresh=(0.0, 0.0, 0.0)):
    """One launch of the segmented reduction -> (level_sums [B,L,5], seg_sums [B,S,6] | None) as CPU fp64 tensors."""
    B, H, W = gt.shape[0], gt.shape[-2], gt.shape[-1]
    dev = pred.device
    S = 0
    seg32 = None
    if seg is not None:
        if seg.dtype.is_floating_point or seg.shape[0] != B or tuple(seg.shape[-2:]) != (H, W):
            raise lib.MvpError(f"segmentation_map must be an integer [B,H,W] map matching the targets, got {seg.dtype} {tuple(seg.shape)}")
        seg32 = seg.to(dev).reshape(B, H, W).to(torch.int32).contiguous()
        lo, hi = (int(v) for v in torch.aminmax(seg32))
        if lo < 0 or hi >= 2048:
            raise lib.MvpError(f"segment ids must lie in [0, 2048) (OneFormer ADE20K ids are 0..149), got [{lo}, {hi}]")
        S = hi + 1
    lv = torch.empty(B, num_levels, 5, dtype=torch.float64, device=dev)
    sg = torch.empty(B, S, 6, dtype=torch.float64, device=dev) if S else None
    ws = torch.empty(int(lib.load().mvp_metrics_breakdown_workspace_bytes(B, num_levels, S)) // 8 + 1, dtype=torch.float64, device=dev)
    a = lib.MetricsBreakdownArgs(lib.ptr(pred), lib.ptr(gt), lib.ptr(seg32), lib.ptr(scale_shift), lib.ptr(lv), lib.ptr(sg), lib.ptr(ws), ws.numel() * 8,
                                 B, H, W, Cp, num_levels, S, thresh[0], thresh[1], thresh[2])
    lib.call("mvp_metrics_breakdown", a)
    return lv.cpu(), (sg.cpu() if S else None)


def _norm(n, mode):
    """depth: num_valid == 0 -> 1e-6 (metrics.py:131-133); snorm: clamp(min=1) (metrics.py:427)."""
    return torch.where(n == 0, torch.full_like(n, 1e-6), n) if mode == "eps" else n.clamp(min=1)


def _finish_breakdown(lv, sg, mode, snorm_group_rmse, is_navi):
    """Bins -> the reference's structures.  ``snorm_group_rmse``: metrics.py:499,513 divide AFTER the square root."""
    from .oneformer_id2label import STUFF, THINGS

    levels = {}
    for i in range(lv.shape[1]):
        n = _norm(lv[:, i, 0], mode)
        levels[f"level_{i + 1}"] = {"d1": (lv[:, i, 1] / n).float(), "d2": (lv[:, i, 2] / n).float(), "d3": (lv[:, i, 3] / n).float(),
                                    "rmse": (lv[:, i, 4] / n).sqrt().float()}
    groups, segments = {}, []
    if not is_navi:
        S = sg.shape[1]
        for name, ids in (("stuff", STUFF), ("things", THINGS)):
            t = sg[:, [i for i in ids if i < S]].sum(dim=1)  # [B, 6]
            n = _norm(t[:, 1], mode)
            for k in range(3):
                groups[f"{name}_d{k + 1}"] = (t[:, 2 + k] / n).float()
            groups[f"{name}_rmse"] = ((t[:, 5].sqrt() / n) if snorm_group_rmse else (t[:, 5] / n).sqrt()).float()
            groups[f"{name}_pixels"] = n.float()
        present = (sg[:, :, 0].sum(dim=0) > 0).nonzero().flatten().tolist()  # torch.unique(segmentation_map): ascending ids
        for sid in present:
            area = _norm(sg[:, sid, 1], mode).float()
            d1 = (sg[:, sid, 2] / _norm(sg[:, sid, 1], mode)).float()
            for b in range(sg.shape[0]):
                segments.append({"segment_id": sid, "image_idx": b, "area": area[b].item(), "d1_ratio": d1[b].item()})
    return groups, levels, segments


def evaluate_depth(depth_pr, depth_gt, segmentation_map=None, image_average=False, scale_invariant=False, nyu_crop=False, num_levels=5, is_navi=False):
    """Reference: metrics.py:106-358.  Returns (global_metrics, metrics_by_level, segment_metrics).  ``is_navi=True`` skips
    the stuff/things and per-segment parts (no segmentation map), exactly as the reference; nyu_crop is forced off there too."""
    out, ss = _depth_metrics(depth_pr, depth_gt, scale_invariant)
    if not is_navi and segmentation_map is None:
        raise ValueError("evaluate_depth(is_navi=False) needs a segmentation_map (metrics.py:180-186)")
    B = depth_pr.shape[0]
    H, W = depth_pr.shape[-2:]
    p = depth_pr.detach().reshape(B, H, W).contiguous().float()
    g = depth_gt.detach().reshape(B, H, W).contiguous().float()
    lv, sg = _breakdown(p, g, None if is_navi else segmentation_map, ss if scale_invariant else None, 0, num_levels)
    out = out.cpu()
    gm = {k: out[:, i] for i, k in enumerate(_DEPTH_KEYS)}
    for k in ("mean_pred", "mean_gt"):  # the reference returns these as [B,1,1]
        gm[k] = gm[k].view(-1, 1, 1)
    groups, levels, segments = _finish_breakdown(lv, sg, "eps", False, is_navi)
    gm.update(groups)
    if image_average:
        gm = {k: v.mean() for k, v in gm.items()}
        levels = {L: {k: v.mean() for k, v in d.items()} for L, d in levels.items()}
    return gm, levels, segments


def evaluate_surface_norm(snorm_pr, snorm_gt, segmentation_map=None, image_average=False, num_levels=5, thresh=[11.25, 22.5, 30.0], is_navi=False):
    """Reference: metrics.py:397-577."""
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
    if not is_navi and segmentation_map is None:
        raise ValueError("evaluate_surface_norm(is_navi=False) needs a segmentation_map (metrics.py:469-475)")
    H, W = snorm_gt.shape[-2:]
    lv, sg = _breakdown(p.view(B, Cp, H, W), g.view(B, 3, H, W), None if is_navi else segmentation_map, None, Cp, num_levels, thresh)
    out = out.cpu()
    gm = {k: out[:, i] for i, k in enumerate(["d1", "d2", "d3", "rmse"])}
    groups, levels, segments = _finish_breakdown(lv, sg, "one", True, is_navi)
    gm.update(groups)
    if image_average:
        gm = {k: v.mean() for k, v in gm.items()}
        levels = {L: {k: v.mean() for k, v in d.items()} for L, d in levels.items()}
    return gm, levels, segments
