"""evals/models/utils.py surface used by the hot path (center_padding, tokens_to_output,
sincos pos-embed).  Pure shape/host helpers; tensor math here is O(B*C) glue."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from mvp.backbone import sincos_pos_embed_2d


def center_padding(images, patch_size):
    """Reference: evals/models/utils.py:55-72 (note: a non-ragged dim still receives a full
    patch of padding when the other dim is ragged).  The HIP backbone applies the same
    padding inside its patch-gather kernel; this helper exists for API parity."""
    _, _, h, w = images.shape
    dh, dw = h % patch_size, w % patch_size
    if dh == 0 and dw == 0:
        return images
    ph, pw = patch_size - dh, patch_size - dw
    return F.pad(images, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))


def get_2d_sincos_pos_embed(embed_dim, grid_size, add_cls_token=False):
    """Reference: evals/models/utils.py:75-102."""
    return sincos_pos_embed_2d(embed_dim, grid_size, add_cls_token)


def tokens_to_output(output_type, dense_tokens, cls_token, feat_hw):
    """Reference: evals/models/utils.py:105-124 (host-side view/permute glue)."""
    if output_type == "cls":
        assert cls_token is not None
        return cls_token
    if output_type == "gap":
        return dense_tokens.mean(dim=1)
    h, w = feat_hw
    b, _, c = dense_tokens.shape
    grid = dense_tokens.reshape(b, h, w, c).permute(0, 3, 1, 2)
    if output_type == "dense":
        return grid.contiguous()
    if output_type == "dense-cls":
        assert cls_token is not None
        return torch.cat((grid, cls_token[:, :, None, None].repeat(1, 1, h, w)), dim=1).contiguous()
    raise ValueError()
