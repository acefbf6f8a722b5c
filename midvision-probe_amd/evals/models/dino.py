"""evals.models.dino.DINO — drop-in for the reference wrapper (evals/models/dino.py:9-210),
ViT-B/16 dense (multi-layer) feature extraction on the HIP kernels."""
from __future__ import annotations

import warnings

import torch
import torch.nn as nn

from mvp import backbone as bb


class DINO(bb.ViTBackbone):
    def __init__(self, dino_name="dino", model_name="vitb16", output="dense", layer=-1, return_multilayer=False, add_norm=False,
                 return_kqv=False, fixed_size=480, mode_selected="k", return_layers=None, return_cls=False,
                 weights=None, precision=None, init_seed=0):
        super().__init__()
        feat_dims = {"vitb8": 768, "vitb16": 768, "vitb14": 768, "vitb14_reg": 768, "vitl14": 1024, "vitg14": 1536}
        if dino_name != "dino" or model_name != "vitb16":
            raise NotImplementedError("the HIP path covers DINO ViT-B/16 (configs/backbone/dino_b16.yaml)")
        if return_kqv and mode_selected not in ("k", "q", "v", "kqv"):
            raise ValueError(f"mode_selected {mode_selected!r}: one of k, q, v, kqv (dino.py:126-139)")
        self.arch = "vit"
        self.return_cls = return_cls
        self.dino_name, self.model_name = dino_name, model_name
        self.checkpoint_name = f"{dino_name}_{model_name}"
        # reference: torch.hub.load("facebookresearch/dino", ...) (dino.py:40) — no network here:
        # local checkpoint (MVP_CKPT_DIR/<checkpoint_name>.pth) or an explicit state dict, else seeded random init.
        sd = weights
        if sd is None:
            path = bb.find_checkpoint(self.checkpoint_name, "dino_vitbase16_pretrain")
            if path is not None:
                sd = bb.load_checkpoint_file(path)
            else:
                warnings.warn(f"no local checkpoint for {self.checkpoint_name}: using seeded random init (seed={init_seed})")
                sd = bb.random_vit_state_dict(seed=init_seed)
        self.vit = bb.ViTParams(sd).eval()
        self.has_registers = "_reg" in model_name
        self.patch_size = self.vit.patch_embed.proj.weight.shape[-1]
        assert output in ["cls", "gap", "dense", "dense-cls"]
        self.output = output
        feat_dim = self.vit.embed_dim  # == feat_dims[model_name] for real checkpoints
        feat_dim = feat_dim * 2 if output == "dense-cls" else feat_dim
        self._setup_taps(feat_dim, layer, return_multilayer, add_norm, self.vit.depth)
        if output == "dense-cls":
            feat_dim = feat_dim // 2
        self.batchnorms = nn.ModuleList([nn.BatchNorm1d(feat_dim) for _ in self.multilayers])
        self.return_kqv, self.fixed_size, self.mode_selected = return_kqv, fixed_size, mode_selected
        self.heads, self.ln_eps, self.pos_embed_mode = self.vit.embed_dim // 64, 1e-6, "dino"
        self.set_precision(precision or bb.default_precision())

    def forward(self, images):
        if self.return_kqv:  # dino.py:164-169
            return self.extract_kqv(self.preprocess_image(images)[0])
        if len(self.multilayers) == 1 and self.return_cls:
            # dino.py:206-207: embeds[0][:, 0] — the (tap-BN normalised when add_norm) CLS token of the single tap
            return self._extract(images, want_cls=True).cls[0]
        taps = self._extract(images)
        return self._finish(taps)
