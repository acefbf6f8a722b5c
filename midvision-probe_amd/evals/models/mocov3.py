"""evals.models.mocov3.MoCoV3 — drop-in for evals/models/mocov3.py:20-186 (timm ViT-B/16,
input force-resized to 224^2, fixed pos-embed, always dense NCHW output)."""
from __future__ import annotations

import warnings

import torch
import torch.nn as nn

from mvp import backbone as bb
from mvp import functional as MF
from mvp import pipeline


class MoCoV3(bb.ViTBackbone):
    params_attr = "model"

    def __init__(self, model_name="vitb16", layer=-1, arch="vitb16", output="dense", return_multilayer=False, add_norm=False,
                 return_kqv=False, fixed_size=480, mode_selected="k", return_cls=False, weights=None, precision=None, init_seed=0):
        super().__init__()
        self.arch = "vit"
        self.return_cls = return_cls
        assert arch == "vitb16", f"Invalid arch: {arch}"
        if return_kqv:
            raise NotImplementedError("return_kqv is outside the hot path")
        sd = weights
        if sd is None:  # reference: wget/gdown download + prepare_state_dict (mocov3.py:70-81)
            path = bb.find_checkpoint("mocov3_vitb16")
            if path is not None:
                raw = bb.load_checkpoint_file(path)
                sd = {k[len("module.base_encoder."):]: v for k, v in raw.items() if k.startswith("module.base_encoder.")}
                sd = {k: v for k, v in sd.items() if not k.startswith("head.")}
            else:
                warnings.warn(f"no local checkpoint for mocov3_vitb16: using seeded random init (seed={init_seed})")
                sd = bb.random_vit_state_dict(seed=init_seed)
        self.model = bb.ViTParams(sd).eval()
        self.output = output
        self.checkpoint_name = f"$mocov3$_{arch}_{output}"
        self.patch_size = self.model.patch_embed.proj.weight.shape[-1]
        self._setup_taps(self.model.embed_dim, -1, return_multilayer, add_norm, self.model.depth)  # 768 for the real ViT-B/16
        self.batchnorms = nn.ModuleList([nn.BatchNorm1d(self.model.embed_dim) for _ in self.multilayers])
        self.return_kqv, self.fixed_size, self.mode_selected = return_kqv, fixed_size, mode_selected
        self.heads, self.ln_eps, self.pos_embed_mode = self.model.embed_dim // 64, 1e-6, "fixed"
        self.set_precision(precision or bb.default_precision())

    def forward(self, images):
        with torch.no_grad():
            images = MF.interpolate(images, size=(224, 224), mode="bilinear", align_corners=False)  # mocov3.py:150-152
        if len(self.multilayers) == 1 and self.return_cls:
            return self.engine().forward_tokens(images, self.multilayers[0] + 1)[:, 0]
        taps = self._extract(images)
        if isinstance(taps, bb.TapGroups):  # several batches stacked into one forward (mvp/pipeline.py): one result per batch
            return pipeline.GroupedFeatures((t[0] if len(t) == 1 else t) for t in taps)
        return taps[0] if len(taps) == 1 else taps
