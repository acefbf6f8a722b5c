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
    images, kps_i, kps_j, thresh_scale = _pair_inputs(instance)
    feats = model(images)  # NB: the reference runs this outside no_grad with the wrapper in train mode
    if isinstance(feats, list):
        feats = torch.cat(feats, dim=1)
    res = correspondence(feats[0], feats[1], kps_i[:, :2], return_heatmaps=return_heatmaps)
    out = _pck_rows(res[0].float().cpu() / feats.shape[-1], kps_i, kps_j, thresh_scale)
    if return_heatmaps:
        return (*out, res[2])
    return out


class _PinnedRing:
    """Host -> device staging through a few reused pinned buffers: the copy engine moves them asynchronously, where a pageable
    ``.to(device)`` blocks the host until the copy has run (and with it the whole look-ahead of the pipeline)."""

    def __init__(self, slots: int):
        self.slots, self.k = [dict() for _ in range(max(2, slots))], 0

    def to_device(self, parts, dev, dtype=torch.float32) -> torch.Tensor:
        """torch.stack(parts) on the device."""
        slot = self.slots[self.k % len(self.slots)]
        self.k += 1
        shape = (len(parts),) + tuple(parts[0].shape)
        ev = slot.get("event")
        if ev is not None:
            ev.synchronize()  # the copy that last read this pinned buffer has finished (normally long ago)
        buf = slot.get("buf")
        if buf is None or tuple(buf.shape) != shape or buf.dtype != dtype:
            buf = slot["buf"] = torch.empty(shape, dtype=dtype).pin_memory()
        torch.stack([p.to(dtype) for p in parts], out=buf)
        out = torch.empty(shape, dtype=dtype, device=dev)
        out.copy_(buf, non_blocking=True)
        slot["event"] = torch.cuda.Event()
        slot["event"].record()
        return out


def _pair_inputs(instance, ring: "_PinnedRing" = None):
    """(images [2,3,S,S] on the device, kps_i, kps_j with (x, y) scaled to [0,1], thresh_scale): evaluate_spair_correspondence.py:47-60."""
    img_i, mask_i, kps_i, img_j, mask_j, kps_j, thresh_scale, _ = instance
    dev = torch.device("cuda", torch.cuda.current_device())
    images = torch.stack((img_i, img_j)).to(dev) if ring is None else ring.to_device((img_i, img_j), dev)
    assert images.shape[-1] == images.shape[-2], "assuming square images here"
    kps_i = kps_i.float().clone()
    kps_j = kps_j.float().clone()
    kps_i[:, :2] = kps_i[:, :2] / images.shape[-1]
    kps_j[:, :2] = kps_j[:, :2] / images.shape[-1]
    return images, kps_i, kps_j, thresh_scale


def _pck_rows(pred_kp: torch.Tensor, kps_i: torch.Tensor, kps_j: torch.Tensor, thresh_scale):
    """Host-side PCK bookkeeping of one pair (evaluate_spair_correspondence.py:86-98) from the predicted keypoints in [0,1]."""
    errors = (pred_kp[:, None, :] - kps_j[None, :, :2]).norm(p=2, dim=-1)
    errors = errors / thresh_scale
    valid_kps = (kps_i[:, None, 2] * kps_j[None, :, 2]) == 1
    in_both = valid_kps.diagonal()
    errors[valid_kps.logical_not()] = 1e3
    error_same = errors.diagonal()[in_both]
    error_nn, index_nn = errors[in_both].min(dim=1)
    index_same = in_both.nonzero().squeeze(1)
    return error_same, error_nn, index_same, index_nn


def shard_pairs(n_pairs: int, rank: int, world: int):
    """Image pairs are independent (tap BN couples only the two images of a pair): rank r takes
    pairs r, r+W, ...; per-pair error vectors are gathered at the end."""
    return list(range(rank, n_pairs, world))


def evaluate_dataset(model, dataset, thresh, verbose=False, rank: int = 0, world: int = 1):
    """Reference: evaluate_spair_correspondence.py:104-121 -> (recall in %, confusion matrix).  With world > 1 the pairs are sharded
    (rank r takes pairs r, r + W, ...; pairs are independent) and the per-pair error / index vectors are gathered with
    one all_gather_object at the end (host-side lists of a few floats per pair), so every rank returns the full-dataset result."""
    idx = shard_pairs(len(dataset), rank, world)
    # The pairs' forwards are kept in flight (mvp/pipeline.py; pairs are independent, the wrapper's train-mode tap BN sees one pair at a
    # time and its running statistics are updated in pair order), the correspondence kernel of pair t runs under the forwards of the
    # next pairs, and the predicted keypoints stay on the device until the end: one host sync per shard instead of one per pair.
    from .pipeline import pipelined_features

    ring = _PinnedRing(8)  # images and keypoints go through pinned staging: nothing in the loop blocks the host

    def pairs():
        for i in idx:
            images, kps_i, kps_j, thresh_scale = _pair_inputs(dataset[i], ring)
            kp_dev = ring.to_device((kps_i[:, :2].contiguous(),), images.device)[0]
            yield {"image": images, "meta": (i, kps_i, kps_j, thresh_scale, kp_dev)}

    # Jobs with more than one rank launch their forwards eagerly by default (mvp/pipeline.py: a lazy graph capture is a device-wide sync
    # that must not land beside a pending all-reduce).  This loop has NO collective in flight — the one exchange is the
    # all_gather_object after it — so it keeps hipGraph replay at any world size (eager launches: 48-80 pairs/s against 115-148 at
    # 800^2, profiles/r04_fullsize_configs.txt); MVP_PIPELINE_GRAPHS still wins when set.
    import os

    graphs = True if (world > 1 and os.environ.get("MVP_PIPELINE_GRAPHS") is None) else None
    pending = []
    for b, feats in pipelined_features(model, pairs(), graphs=graphs):
        f = torch.cat(list(feats), dim=1) if isinstance(feats, (list, tuple)) else feats
        i, kps_i, kps_j, thresh_scale, kp_dev = b["meta"]
        pred_xy, _ = correspondence(f[0], f[1], kp_dev)
        pending.append((i, pred_xy, f.shape[-1], kps_i, kps_j, thresh_scale))
    outs = [(i,) + _pck_rows(pred_xy.float().cpu() / fw, kps_i, kps_j, ts) for i, pred_xy, fw, kps_i, kps_j, ts in pending]
    if world > 1:
        import torch.distributed as dist

        gathered = [None] * world
        dist.all_gather_object(gathered, outs)
        outs = sorted((o for part in gathered for o in part), key=lambda o: o[0])  # dataset order, as the reference's single loop
    errors = torch.cat([o[1] for o in outs])
    src_ind = torch.cat([o[3] for o in outs])
    tgt_ind = torch.cat([o[4] for o in outs])
    kp_max = int(max(src_ind.max(), tgt_ind.max())) + 1
    confusion = torch.zeros((kp_max, kp_max))
    for src, tgt in torch.stack((src_ind, tgt_ind), dim=1):
        confusion[src, tgt] += 1
    recall = (errors < thresh).float().mean().item() * 100.0
    return recall, confusion


class SyntheticSPair(torch.utils.data.Dataset):
    """SPair-71k-shaped instances (evals/datasets/spair.py contract consumed by compute_errors): (img_i, mask_i, kps_i [K,3],
    img_j, mask_j, kps_j [K,3], thresh_scale, meta).  img_j is img_i shifted by a whole number of patches, so that a feature
    extractor with any spatial sensitivity can solve it: used by the tests and the config-#5 entry script (no dataset files in scope)."""

    def __init__(self, num_pairs=8, image_size=800, num_kps=20, seed=0, patch=16):
        self.n, self.S, self.K, self.seed, self.patch = num_pairs, image_size, num_kps, seed, patch

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        import numpy as np

        g = torch.Generator().manual_seed(self.seed * 7919 + i)
        S, P = self.S, self.patch
        img_i = torch.randn(3, S, S, generator=g)
        dy, dx = (int(v) for v in torch.randint(-2, 3, (2,), generator=g))
        img_j = torch.roll(img_i, shifts=(dy * P, dx * P), dims=(1, 2))
        margin = 3 * P
        kp = torch.randint(margin, S - margin, (self.K, 2), generator=g).float()  # (x, y)
        vis = (torch.rand(self.K, 1, generator=g) > 0.1).float()
        kps_i = torch.cat([kp, vis], 1)
        kps_j = torch.cat([kp + torch.tensor([dx * P, dy * P], dtype=torch.float32), vis], 1)
        mask = np.ones((S, S))
        return img_i, mask, kps_i, img_j, mask, kps_j, 1.0, {"pair": i}
