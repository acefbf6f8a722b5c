"""evals.models.ibot.iBOT — drop-in for evals/models/ibot.py:15-220 (ViT-B/16)."""
from __future__ import annotations

import warnings

import torch
import torch.nn as nn

from mvp import backbone as bb


class iBOT(bb.ViTBackbone):
    def __init__(self, model_type="base", output="dense", layer=-1, return_multilayer=False, add_norm=False, return_kqv=False,
                 fixed_size=480, mode_selected="k", return_cls=False, weights=None, precision=None, init_seed=0):
        super().__init__()
        self.arch = "vit"
        self.return_cls = return_cls
        assert output in ["gap", "dense", "cls", "dense-cls"]
        self.output = output
        self.return_multilayer = return_multilayer
        model_dict = {"base": "ibot_vitb16", "base_in22k": "ibot_vitb16_in22k"}
        if model_type not in model_dict:
            raise NotImplementedError("the HIP path covers iBOT ViT-B/16 (model_type base / base_in22k)")
        if return_kqv and mode_selected not in ("k", "q", "v", "kqv"):
            raise ValueError(f"mode_selected {mode_selected!r}: one of k, q, v, kqv (ibot.py:169-180)")
        ckpt_name = model_dict[model_type]
        sd = weights
        if sd is None:  # reference: urlretrieve + torch.load(...)["state_dict"] (ibot.py:46-56)
            path = bb.find_checkpoint(ckpt_name)
            if path is not None:
                sd = {k.replace("module.", ""): v for k, v in bb.load_checkpoint_file(path).items()}
                sd = {k: v for k, v in sd.items() if not k.startswith("head.")}
            else:
                warnings.warn(f"no local checkpoint for {ckpt_name}: using seeded random init (seed={init_seed})")
                sd = bb.random_vit_state_dict(seed=init_seed)
        self.vit = bb.ViTParams(sd).eval()
        self.patch_size = 16
        self.checkpoint_name = "$ibot$" + ckpt_name
        feat_dim = self.vit.embed_dim * (2 if output == "dense-cls" else 1)
        self._setup_taps(feat_dim, layer, return_multilayer, add_norm, self.vit.depth)
        self.batchnorms = nn.ModuleList([nn.BatchNorm1d(feat_dim) for _ in self.multilayers])
        self.return_kqv, self.fixed_size, self.mode_selected = return_kqv, fixed_size, mode_selected
        self.heads, self.ln_eps, self.pos_embed_mode = self.vit.embed_dim // 64, 1e-6, "dino"
        self.set_precision(precision or bb.default_precision())

    def forward(self, images):
        if self.return_kqv:  # ibot.py:182-186
            return self.extract_kqv(self.preprocess_image(images)[0])
        if len(self.multilayers) == 1 and self.return_cls:
            # ibot.py:199-200: raw CLS token of the tapped block (no BN, no later blocks)
            eng = self.engine()
            with torch.no_grad():
                return eng.forward_tokens(images, self.multilayers[0] + 1)[:, 0]
        return self._finish(self._extract(images))
