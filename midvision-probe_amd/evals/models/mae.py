"""evals.models.mae.MAE — drop-in for evals/models/mae.py:11-237 (HF ViT-MAE encoder, no
masking; sincos pos-embed rebuilt per image size; LayerNorm eps 1e-12; taps are HF
``hidden_states[i]`` = the INPUT of block i, quirk Q4)."""
from __future__ import annotations

import warnings

import torch
import torch.nn as nn

from mvp import backbone as bb


class MAE(bb.ViTBackbone):
    tap_input_of_block = True

    def __init__(self, checkpoint="facebook/vit-mae-base", output="dense", layer=-1, return_multilayer=False, add_norm=False,
                 return_kqv=False, fixed_size=480, mode_selected="k", return_cls=False, weights=None, precision=None, init_seed=0):
        super().__init__()
        self.arch = "vit"
        self.return_cls = return_cls
        assert output in ["cls", "gap", "dense"], "Options: [cls, gap, dense]"
        self.output = output
        if return_kqv:
            raise NotImplementedError("return_kqv is outside the hot path")
        self.checkpoint_name = "$mae$" + checkpoint.split("/")[1]
        sd = weights
        if sd is None:  # reference: ViTMAEForPreTraining.from_pretrained(checkpoint).vit (mae.py:33)
            path = bb.find_checkpoint(checkpoint.split("/")[1], "vit-mae-base")
            if path is not None:
                sd = bb.load_checkpoint_file(path)
            else:
                warnings.warn(f"no local checkpoint for {checkpoint}: using seeded random init (seed={init_seed})")
                sd = bb.random_vit_state_dict(seed=init_seed)
                sd["pos_embed"] = torch.from_numpy(bb.sincos_pos_embed_2d(768, (14, 14), True)).float().unsqueeze(0)
        if any("encoder.layer.0." in k for k in sd):
            sd = bb.hf_vitmae_to_fused(sd)
        self.vit = bb.ViTParams(sd).eval()
        self.patch_size = self.vit.patch_embed.proj.weight.shape[-1]
        self.image_size = (224, 224)
        self.feat_h = self.image_size[0] // self.patch_size
        self.feat_w = self.image_size[1] // self.patch_size
        self._setup_taps(self.vit.embed_dim, layer, return_multilayer, add_norm, self.vit.depth)
        self.batchnorms = nn.ModuleList([nn.BatchNorm1d(self.vit.embed_dim) for _ in self.multilayers])
        self.return_kqv, self.fixed_size, self.mode_selected = return_kqv, fixed_size, mode_selected
        self.heads, self.ln_eps, self.pos_embed_mode = self.vit.embed_dim // 64, 1e-12, "fixed"
        self.set_precision(precision or bb.default_precision())

    def resize_pos_embed(self, image_size):
        """mae.py:74-89: rebuild the 2-D sincos table for the new grid."""
        assert image_size[0] % self.patch_size == 0
        assert image_size[1] % self.patch_size == 0
        self.feat_h = image_size[0] // self.patch_size
        self.feat_w = image_size[1] // self.patch_size
        self.image_size = tuple(image_size)
        pe = bb.sincos_pos_embed_2d(self.vit.embed_dim, (self.feat_h, self.feat_w), add_cls_token=True)
        dev = self.vit.pos_embed.device
        self.vit.pos_embed = nn.Parameter(torch.from_numpy(pe).float().unsqueeze(0).to(dev), requires_grad=False)

    def forward(self, images):
        if tuple(self.image_size) != tuple(images.shape[-2:]):
            self.resize_pos_embed(images.shape[-2:])
        if len(self.multilayers) == 1 and self.return_cls:
            return self.engine().forward_tokens(images, self.multilayers[0])[:, 0]
        return self._finish(self._extract(images))
