"""Shared base of the ResNet-50 wrappers (evals/models/dino_res50.py, mocov3_res50.py and the other
SSL ResNet-50 checkpoints that use the same template)."""
from __future__ import annotations

import math
import warnings
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import backbone as bb
from . import functional as MF
from . import lib
from .resnet import LAYERS, ResNetEngine
from .vit import parse_precision

WIDTHS = (64, 128, 256, 512)


def random_resnet50_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    """torchvision resnet50(weights=None) statistics: kaiming-normal(fan_out) convs, BN weight 1 /
    bias 0, running mean 0 / var 1 (util.py:27-51 builds exactly that before loading a checkpoint)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def conv(name, cout, cin, k):
        sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * math.sqrt(2.0 / (cout * k * k))

    def bn(name, c):
        sd[name + ".weight"], sd[name + ".bias"] = torch.ones(c), torch.zeros(c)
        sd[name + ".running_mean"], sd[name + ".running_var"] = torch.zeros(c), torch.ones(c)

    conv("conv1", 64, 3, 7); bn("bn1", 64)
    inp = 64
    for li, (n, wd) in enumerate(zip(LAYERS, WIDTHS), start=1):
        for bi in range(n):
            p = f"layer{li}.{bi}."
            conv(p + "conv1", wd, inp, 1); bn(p + "bn1", wd)
            conv(p + "conv2", wd, wd, 3); bn(p + "bn2", wd)
            conv(p + "conv3", wd * 4, wd, 1); bn(p + "bn3", wd * 4)
            if bi == 0:
                conv(p + "downsample.0", wd * 4, inp, 1); bn(p + "downsample.1", wd * 4)
            inp = wd * 4
    return sd


class ResNetParams(nn.Module):
    """Parameter / buffer container with torchvision's resnet50 key layout."""

    def __init__(self, sd: Dict[str, torch.Tensor]):
        super().__init__()
        for k, v in sd.items():
            if not torch.is_tensor(v) or k.startswith("fc."):
                continue
            parts = k.split(".")
            mod = self
            for p in parts[:-1]:
                if not hasattr(mod, p):
                    setattr(mod, p, bb._Named())
                mod = getattr(mod, p)
            if parts[-1] in ("running_mean", "running_var", "num_batches_tracked"):
                mod.register_buffer(parts[-1], v.clone())
            else:
                mod.register_parameter(parts[-1], nn.Parameter(v.clone().float(), requires_grad=False))


class ResNetBackbone(nn.Module):
    feat_dims_all = [(64, 240), (256, 120), (512, 60), (1024, 30), (2048, 15)]  # nominal (quirk Q8)
    supports_pipelining = True  # the engine allocates per forward on the launching stream and defers the tap-BN running-statistics updates to the consumer (mvp/pipeline.py)
    graph_safe = True  # per-forward buffers come from the capturing graph's private pool: a slot's forward replays on fixed addresses

    def _setup(self, sd, output, return_layers, return_multilayer, add_norm, fixed_size, precision):
        self.model = ResNetParams(sd).eval()
        self.output = output
        self.return_layers = return_layers if return_layers is not None else [0, 1, 2, 3, 4]
        self.feat_dims = list(self.feat_dims_all)
        feat_dims = [self.feat_dims[i] for i in self.return_layers]
        self.patch_size = 0
        if return_multilayer:
            self.feat_dim = feat_dims
            self.multilayers = self.return_layers
        else:
            self.feat_dim = feat_dims[-1]
            self.multilayers = [self.return_layers[-1]]
        self.layer = "-".join(str(_x) for _x in self.multilayers)
        self.batchnorms = nn.ModuleList([nn.BatchNorm2d(fd[0]) for fd in self.feat_dims])
        self.add_norm = add_norm
        self.fixed_size = fixed_size
        self._precision = parse_precision(precision or bb.default_precision())

    def set_precision(self, precision):
        self._precision = parse_precision(precision)

    def engine(self) -> ResNetEngine:
        sig = tuple((p.data_ptr(), p._version) for p in self.model.parameters()) + (self._precision,)
        if getattr(self, "_engine_sig", None) != sig:
            dev = next(self.model.parameters()).device
            if dev.type != "cuda":
                raise lib.MvpError("backbone parameters are on the CPU: call model.to('cuda') — the HIP path has no CPU fallback")
            self._engine_obj = ResNetEngine(self.model.state_dict(), precision=self._precision, device=dev)
            self._engine_sig = sig
        return self._engine_obj

    def _resize(self, x):
        """Resize((fixed_size, fixed_size)), dino_res50.py:80,85 (torchvision resizes tensors with antialias=True).  Identity or
        up-sampling on both axes is plain bilinear (the antialias filter degenerates to it); if an axis down-samples (NYU
        480x640 with center_crop False -> 480x480) the stretched triangle filter of ATen's _upsample_bilinear2d_aa runs."""
        S = self.fixed_size
        H, W = x.shape[-2:]
        if (H, W) == (S, S):
            return x
        if H > S or W > S:
            return MF.resize_antialias(x, (S, S))
        return MF.interpolate(x, size=(S, S), mode="bilinear", align_corners=False)

    def forward(self, x):
        if not x.is_cuda:
            raise lib.MvpError("images must be on the HIP device (no CPU fallback)")
        with torch.no_grad():
            x = self._resize(x)
            bns, mode = None, 2
            if self.add_norm:
                bns = [dict(weight=bn.weight, bias=bn.bias, running_mean=bn.running_mean, running_var=bn.running_var,
                            num_batches_tracked=bn.num_batches_tracked) for bn in self.batchnorms]
                mode = 0 if self.training else 1
            outs = self.engine().forward_taps(x, self.multilayers, bn=bns, bn_mode=mode)
        return outs[0] if len(outs) == 1 else outs


def make_ssl_resnet50(class_name: str, tag: str, prefixes, local_names, ref: str):
    """Factory for the ResNet-50 SSL wrappers that share the reference's template
    (arch, return_layers, output, return_multilayer, add_norm, return_kqv, fixed_size, mode_selected, return_cls):
    they differ only in the checkpoint-name tag and in the state-dict prefix to strip."""

    def __init__(self, arch="resnet50", return_layers=None, output="dense", return_multilayer=False, add_norm=False, return_kqv=False,
                 fixed_size=480, mode_selected="k", return_cls=False, weights=None, precision=None, init_seed=0):
        ResNetBackbone.__init__(self)
        assert arch == "resnet50", f"Invalid arch: {arch}"
        if return_kqv:
            raise NotImplementedError("return_kqv is outside the hot path")
        self.arch = arch
        self.return_cls = return_cls
        sd = weights
        if sd is None:
            path = bb.find_checkpoint(*local_names)
            if path is not None:
                sd = bb.load_checkpoint_file(path)
                for pre in prefixes:  # prepare_state_dict(remove_prefix=...), util.py:106-120
                    if any(k.startswith(pre) for k in sd):
                        sd = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
                sd = {k: v for k, v in sd.items() if not (k.startswith("fc.") or k.startswith("head."))}
            else:
                warnings.warn(f"no local checkpoint for {class_name}: using seeded random init (seed={init_seed})")
                sd = random_resnet50_state_dict(init_seed)
        self._setup(sd, output, return_layers, return_multilayer, add_norm, fixed_size, precision)
        self.checkpoint_name = f"{tag}_{arch}_{output}_{self.return_layers}"
        self.return_kqv, self.mode_selected = return_kqv, mode_selected

    return type(class_name, (ResNetBackbone,), {"__init__": __init__, "__doc__": f"Drop-in for {ref} (same ResNet-50 template as mocov3_res50.py)."})
