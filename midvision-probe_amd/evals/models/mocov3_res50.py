"""evals.models.mocov3_res50.MoCoV3_RES — drop-in for evals/models/mocov3_res50.py:14-116."""
from __future__ import annotations

import warnings

from mvp import backbone as bb
from mvp.resnet_backbone import ResNetBackbone, random_resnet50_state_dict


class MoCoV3_RES(ResNetBackbone):
    def __init__(self, arch="resnet50", return_layers=None, output="dense", return_multilayer=False, add_norm=False, return_kqv=False,
                 fixed_size=480, mode_selected="k", return_cls=False, weights=None, precision=None, init_seed=0):
        super().__init__()
        assert arch == "resnet50", f"Invalid arch: {arch}"
        if return_kqv:
            raise NotImplementedError("return_kqv is outside the hot path")
        self.arch = arch
        self.return_cls = return_cls
        sd = weights
        if sd is None:  # reference: wget + prepare_state_dict(remove "module.base_encoder.") (mocov3_res50.py:83-95)
            path = bb.find_checkpoint("mocov3_resnet50", "r-50-1000ep")
            if path is not None:
                raw = bb.load_checkpoint_file(path)
                sd = {k[len("module.base_encoder."):]: v for k, v in raw.items() if k.startswith("module.base_encoder.")} or raw
            else:
                warnings.warn(f"no local checkpoint for mocov3 resnet50: using seeded random init (seed={init_seed})")
                sd = random_resnet50_state_dict(init_seed)
        self._setup(sd, output, return_layers, return_multilayer, add_norm, fixed_size, precision)
        self.checkpoint_name = f"$mocov3$_{arch}_{output}_{self.return_layers}"
        self.return_kqv, self.mode_selected = return_kqv, mode_selected
