"""SPair-71k keypoint-correspondence hot path (evaluate_spair_correspondence.py:45-103).

``correspondence`` runs the fused HIP kernel (L2-normalise over C, bilinear keypoint sampling,
cosine heat-map, 2-D argmax); ``compute_errors`` mirrors the reference function's signature and
return values (the PCK bookkeeping on K<=~30 keypoints is host-side glue)."""
from __future__ import annotations

import numpy as np
import torch

from . import lib, ops


def argmax_2d(x: torch.Tensor, max_value: bool = True) -> torch.Tensor:
    """Reference: evals/utils/correspondence.py:179-190 — x [..., h, w] -> [..., 2] int64 (col, row) of the flat argmax
    (argmin when max_value=False); ties resolve to the lowest flat index, as torch."""
    if not x.is_cuda:
        raise lib.MvpError("argmax_2d needs a device tensor (no CPU fallback)")
    h, w = x.shape[-2:]
    xs = x.detach().reshape(-1, h, w).contiguous().float()
    out = torch.empty(xs.shape[0], 2, dtype=torch.int64, device=x.device)
    lib.call("mvp_argmax_2d", lib.Argmax2dArgs(lib.ptr(xs), lib.ptr(out), xs.shape[0], h, w, int(bool(max_value))))
    return out.view(*x.shape[:-2], 2)


def correspondence(feats_i: torch.Tensor, feats_j: torch.Tensor, kps_i_xy01: torch.Tensor, return_heatmaps: bool = False):
    """feats_* [C,h,w] fp32 device maps (un-normalised), kps_i_xy01 [K,2] in [0,1] (x,y).
    Returns (pred_xy [K,2] int64 (col,row), max_val [K]) on the device (+ heat-maps [K,h,w] when asked)."""
    if not feats_i.is_cuda:
        raise lib.MvpError("spair.correspondence needs device tensors (no CPU fallback)")
    C, h, w = feats_i.shape
    K = kps_i_xy01.shape[0]
    dev = feats_i.device
    ndc = (kps_i_xy01.to(dev, torch.float32) * 2 - 1).contiguous()
    out_xy = torch.empty(K, 2, dtype=torch.int64, device=dev)
    out_val = torch.empty(K, dtype=torch.float32, device=dev)
    ws = torch.empty((int(lib.load().mvp_corr_workspace_bytes(C, h, w, K)) + 3) // 4, dtype=torch.float32, device=dev)
    heat = torch.empty(K, h, w, dtype=torch.float32, device=dev) if return_heatmaps else None
    ops.corr_argmax(feats_i.contiguous().float(), feats_j.contiguous().float(), ndc, out_xy, out_val, ws, C, h, w, K, heat_out=heat)
    return (out_xy, out_val, heat) if return_heatmaps else (out_xy, out_val)


def compute_errors(model, instance, mask_feats=False, return_heatmaps=False):
    """Reference: evaluate_spair_correspondence.py:45-103.  ``instance`` =
    (img_i, mask_i, kps_i, img_j, mask_j, kps_j, thresh_scale, _)."""
    if mask_feats:
        # the reference multiplies feats [2,C,h,w] by masks [2,h,w] (evaluate_spair_correspondence.py:61-62): that broadcast only
        # type-checks for C == 2, i.e. it raises for every real backbone; no caller in the reference passes mask_feats=True
        raise NotImplementedError("mask_feats=True does not broadcast in the reference either (feats [2,C,h,w] * masks [2,h,w])")
    img_i, mask_i, kps_i, img_j, mask_j, kps_j, thresh_scale, _ = instance
    dev = torch.device("cuda", torch.cuda.current_device())
    images = torch.stack((img_i, img_j)).to(dev)
    feats = model(images)  # NB: the reference runs this outside no_grad with the wrapper in train mode
    if isinstance(feats, list):
        feats = torch.cat(feats, dim=1)
    assert images.shape[-1] == images.shape[-2], "assuming square images here"
    kps_i = kps_i.float().clone()
    kps_j = kps_j.float().clone()
    kps_i[:, :2] = kps_i[:, :2] / images.shape[-1]
    kps_j[:, :2] = kps_j[:, :2] / images.shape[-1]
    res = correspondence(feats[0], feats[1], kps_i[:, :2], return_heatmaps=return_heatmaps)
    pred_xy = res[0]
    pred_kp = pred_xy.float().cpu() / feats.shape[-1]
    errors = (pred_kp[:, None, :] - kps_j[None, :, :2]).norm(p=2, dim=-1)
    errors = errors / thresh_scale
    valid_kps = (kps_i[:, None, 2] * kps_j[None, :, 2]) == 1
    in_both = valid_kps.diagonal()
    errors[valid_kps.logical_not()] = 1e3
    error_same = errors.diagonal()[in_both]
    error_nn, index_nn = errors[in_both].min(dim=1)
    index_same = in_both.nonzero().squeeze(1)
    if return_heatmaps:
        return error_same, error_nn, index_same, index_nn, res[2]
    return error_same, error_nn, index_same, index_nn


def shard_pairs(n_pairs: int, rank: int, world: int):
    """Image pairs are independent (tap BN couples only the two images of a pair): rank r takes
    pairs r, r+W, ...; per-pair error vectors are gathered at the end."""
    return list(range(rank, n_pairs, world))
