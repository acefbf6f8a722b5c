"""Oracle: SPair-71k correspondence core, fp32 CPU.  Test infrastructure only.

compute_errors, evaluate_spair_correspondence.py:45-103 (feature part) and argmax_2d,
evals/utils/correspondence.py:179-190 (restated; the golden fixture comes from the reference module imported behind a faiss stub)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def argmax_2d(x: torch.Tensor, max_value: bool = True) -> torch.Tensor:
    """correspondence.py:179-190 — flat argmax (argmin when max_value=False) over the last two dims, returned as (col, row).
    Pinned bit-exactly by tests/golden/spair.npz (outputs of the reference function, incl. ties)."""
    h, w = x.shape[-2:]
    flat = torch.flatten(x, start_dim=-2)
    flat = flat.argmax(dim=-1) if max_value else flat.argmin(dim=-1)
    return torch.stack((flat % w, flat // w), dim=-1)


def correspondence(feats: torch.Tensor, kps_i_xy01: torch.Tensor):
    """feats [2,C,h,w] (source, target); kps_i_xy01 [K,2] keypoints of the source in [0,1].
    Returns (pred_xy [K,2] int64 (col,row), heatmaps [K,h,w])."""
    feats = F.normalize(feats, p=2, dim=1)
    ndc = (kps_i_xy01.float() * 2 - 1)[None, None]
    desc = F.grid_sample(feats[0][None], ndc, mode="bilinear", align_corners=True)[0, :, 0].t()  # [K,C]
    heat = torch.einsum("kf,fhw->khw", desc, feats[1])
    return argmax_2d(heat), heat
