"""evals.models.dino_res50.DINO_RESNET — drop-in for evals/models/dino_res50.py:8-101."""
from __future__ import annotations

import warnings

from mvp import backbone as bb
from mvp.resnet_backbone import ResNetBackbone, random_resnet50_state_dict


class DINO_RESNET(ResNetBackbone):
    def __init__(self, dino_name="dino", model_name="resnet50", output="dense", layer=-1, return_multilayer=False, add_norm=False,
                 return_kqv=False, fixed_size=480, mode_selected="k", return_layers=None, return_cls=False,
                 weights=None, precision=None, init_seed=0):
        super().__init__()
        if return_kqv:
            raise NotImplementedError("return_kqv is outside the hot path")
        self.arch = model_name
        self.return_cls = return_cls
        self.model_name = model_name
        self.checkpoint_name = f"{dino_name}_{model_name}"
        sd = weights
        if sd is None:  # reference: torch.hub.load("facebookresearch/dino", "dino_resnet50") (dino_res50.py:28-30)
            path = bb.find_checkpoint(self.checkpoint_name, "dino_resnet50_pretrain")
            if path is not None:
                sd = bb.load_checkpoint_file(path)
            else:
                warnings.warn(f"no local checkpoint for {self.checkpoint_name}: using seeded random init (seed={init_seed})")
                sd = random_resnet50_state_dict(init_seed)
        self._setup(sd, output, return_layers, return_multilayer, add_norm, fixed_size, precision)
        self.return_kqv, self.mode_selected = return_kqv, mode_selected
