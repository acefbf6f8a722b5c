"""Shared machinery of the evals.models.* ViT wrappers: parameter containers with the
reference's state-dict key layout, checkpoint discovery (local files only — there is no
network), lazy construction of the HIP engine, and the tap / tokens_to_output glue.
"""
from __future__ import annotations

import math
import os
import warnings
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import lib, pipeline
from .vit import TapGroups, TapOutputs, ViTEngine, parse_precision


def default_precision() -> str:
    """MVP_PRECISION when set, else 'f16x2': the ViT blocks' GEMMs as two fp16 products over compensated fp16 pairs
    (include/mvp_hip.h, MVP_PREC_F16X2) — the feature error of 'bf16x3' (1.5e-5 ... 2.3e-5 on the reference's ViT-B/16 goldens, contract
    1e-3) at 2/3 of its matrix work; activations must stay within fp16's range (|LayerNorm output|, |attention output|, |GELU(fc1)| <= 65504).
    'bf16x3' (three bf16 products, fp32's exponent range) is what the ResNet trunk and the probes run either way; 'bf16' (one product)
    fails the 1e-3 feature contract."""
    return os.environ.get("MVP_PRECISION", "f16x2")


def checkpoint_dir() -> str:
    return os.environ.get("MVP_CKPT_DIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "checkpoints"))


def find_checkpoint(*names: str) -> Optional[str]:
    for n in names:
        for ext in ("", ".pth", ".pth.tar", ".pt", ".safetensors"):
            p = os.path.join(checkpoint_dir(), n + ext)
            if os.path.isfile(p):
                return p
    return None


def load_checkpoint_file(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file

        return load_file(path)
    obj = torch.load(path, map_location="cpu", weights_only=True)
    for key in ("state_dict", "model", "teacher"):
        if isinstance(obj, dict) and key in obj and isinstance(obj[key], dict):
            obj = obj[key]
    return obj


def random_vit_state_dict(embed_dim=768, depth=12, mlp_ratio=4.0, patch=16, img=224, seed=0, in_chans=3) -> Dict[str, torch.Tensor]:
    """Seeded random init with the reference's statistics (trunc-normal 0.02 linears, zero
    biases, unit LayerNorm: ibot_transformers.py:293-309).  Used when no local checkpoint
    exists (BASELINE configs are random-init / synthetic by construction)."""
    g = torch.Generator().manual_seed(seed)

    def tn(*shape):
        t = torch.empty(*shape)
        torch.nn.init.trunc_normal_(t, std=0.02, a=-2.0, b=2.0, generator=g)
        return t

    hid = int(embed_dim * mlp_ratio)
    sd = {"cls_token": tn(1, 1, embed_dim), "pos_embed": tn(1, (img // patch) ** 2 + 1, embed_dim)}
    bound = 1.0 / math.sqrt(in_chans * patch * patch)
    sd["patch_embed.proj.weight"] = (torch.rand(embed_dim, in_chans, patch, patch, generator=g) * 2 - 1) * bound
    sd["patch_embed.proj.bias"] = (torch.rand(embed_dim, generator=g) * 2 - 1) * bound
    for i in range(depth):
        p = f"blocks.{i}."
        for n in ("norm1", "norm2"):
            sd[p + n + ".weight"], sd[p + n + ".bias"] = torch.ones(embed_dim), torch.zeros(embed_dim)
        sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"] = tn(3 * embed_dim, embed_dim), torch.zeros(3 * embed_dim)
        sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"] = tn(embed_dim, embed_dim), torch.zeros(embed_dim)
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = tn(hid, embed_dim), torch.zeros(hid)
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = tn(embed_dim, hid), torch.zeros(embed_dim)
    sd["norm.weight"], sd["norm.bias"] = torch.ones(embed_dim), torch.zeros(embed_dim)
    return sd


def hf_vitmae_to_fused(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """HF ViTMAEModel keys (embeddings.*, encoder.layer.i.attention.attention.{query,key,value},
    layernorm_before/after, intermediate.dense, output.dense) -> the fused DINO-style layout."""
    out = {}
    pre = "vit." if any(k.startswith("vit.") for k in sd) else ""
    g = lambda k: sd[pre + k]  # noqa: E731
    out["cls_token"] = g("embeddings.cls_token")
    out["pos_embed"] = g("embeddings.position_embeddings")
    out["patch_embed.proj.weight"] = g("embeddings.patch_embeddings.projection.weight")
    out["patch_embed.proj.bias"] = g("embeddings.patch_embeddings.projection.bias")
    i = 0
    while pre + f"encoder.layer.{i}.layernorm_before.weight" in sd:
        s, d = f"encoder.layer.{i}.", f"blocks.{i}."
        out[d + "norm1.weight"], out[d + "norm1.bias"] = g(s + "layernorm_before.weight"), g(s + "layernorm_before.bias")
        out[d + "norm2.weight"], out[d + "norm2.bias"] = g(s + "layernorm_after.weight"), g(s + "layernorm_after.bias")
        out[d + "attn.qkv.weight"] = torch.cat([g(s + f"attention.attention.{n}.weight") for n in ("query", "key", "value")], 0)
        out[d + "attn.qkv.bias"] = torch.cat([g(s + f"attention.attention.{n}.bias") for n in ("query", "key", "value")], 0)
        out[d + "attn.proj.weight"], out[d + "attn.proj.bias"] = g(s + "attention.output.dense.weight"), g(s + "attention.output.dense.bias")
        out[d + "mlp.fc1.weight"], out[d + "mlp.fc1.bias"] = g(s + "intermediate.dense.weight"), g(s + "intermediate.dense.bias")
        out[d + "mlp.fc2.weight"], out[d + "mlp.fc2.bias"] = g(s + "output.dense.weight"), g(s + "output.dense.bias")
        i += 1
    if pre + "layernorm.weight" in sd:
        out["norm.weight"], out["norm.bias"] = g("layernorm.weight"), g("layernorm.bias")
    return out


def sincos_pos_embed_2d(embed_dim: int, grid_hw, add_cls_token: bool = True) -> np.ndarray:
    """evals/models/utils.py:75-102 + HF get_2d_sincos_pos_embed_from_grid (MAE): half of the
    channels encode the w coordinate ("w goes first"), half the h coordinate; each half is
    [sin | cos] of pos * 1/10000^(2i/D)."""
    gh, gw = grid_hw
    grid_h = np.arange(gh, dtype=np.float32)
    grid_w = np.arange(gw, dtype=np.float32)
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape(2, 1, gh, gw)

    def one_dim(d, pos):
        omega = np.arange(d // 2, dtype=float) / (d / 2.0)
        omega = 1.0 / 10000 ** omega
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    emb = np.concatenate([one_dim(embed_dim // 2, grid[0]), one_dim(embed_dim // 2, grid[1])], axis=1)
    if add_cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


class _Named(nn.Module):
    pass


class ViTParams(nn.Module):
    """Parameter container with the key layout of the DINO / iBOT / timm VisionTransformer
    (cls_token, pos_embed, patch_embed.proj, blocks.i.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}, norm)."""

    def __init__(self, sd: Dict[str, torch.Tensor]):
        super().__init__()
        for k, v in sd.items():
            parts = k.split(".")
            mod = self
            for p in parts[:-1]:
                if not hasattr(mod, p):
                    setattr(mod, p, _Named())
                mod = getattr(mod, p)
            mod.register_parameter(parts[-1], nn.Parameter(v.clone().float(), requires_grad=False))
        self.embed_dim = sd["cls_token"].shape[-1]
        self.depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))

    @property
    def blocks(self):
        return [getattr(self._modules["blocks"], str(i)) for i in range(self.depth)]


def multilayer_indices(n: int) -> List[int]:
    return [n // 4 - 1, n // 2 - 1, n // 4 * 3 - 1, n - 1]


class ViTBackbone(nn.Module):
    """Base of the ViT wrappers.  Subclasses set: self.<params_attr> (ViTParams), heads, ln_eps,
    pos_embed_mode, tap_input_of_block; and call _setup_taps()."""

    params_attr = "vit"
    heads = 12
    ln_eps = 1e-6
    pos_embed_mode = "dino"
    tap_input_of_block = False
    supports_pipelining = True  # per-slot buffers, tap-BN running-statistics updates deferred to the consumer (mvp/pipeline.py)
    graph_safe = True  # a pipelined forward launches only this library's kernels on fixed buffers: it can be captured in a hipGraph

    def supports_grouping(self) -> bool:
        """True when the pipeline may stack several batches into ONE forward (mvp/pipeline.py, ``group``): the dense / multilayer
        paths, whose ``forward`` ends in ``_finish(_extract(images))``.  The single-tap ``return_cls`` shortcuts return a bare
        token tensor and stay one batch per forward."""
        return not (len(self.multilayers) == 1 and getattr(self, "return_cls", False))

    def _setup_taps(self, feat_dim, layer, return_multilayer, add_norm, num_layers):
        multilayers = multilayer_indices(num_layers)
        if return_multilayer:
            self.feat_dim = [feat_dim] * 4
            self.multilayers = multilayers
        else:
            self.feat_dim = feat_dim
            self.multilayers = [multilayers[-1] if layer == -1 else layer]
        self.layer = "-".join(str(x) for x in self.multilayers)
        self.add_norm = add_norm

    # ---- engine management
    def _params(self) -> ViTParams:
        return getattr(self, self.params_attr)

    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self._params().parameters()) + (self._precision,)

    def engine(self) -> ViTEngine:
        sig = self._signature()
        if getattr(self, "_engine_sig", None) != sig:
            params = self._params()
            dev = next(params.parameters()).device
            if dev.type != "cuda":
                raise lib.MvpError("backbone parameters are on the CPU: call model.to('cuda') — the HIP path has no CPU fallback")
            sd = {k: v for k, v in params.state_dict().items()}
            self._engine_obj = ViTEngine(sd, heads=self.heads, patch=self.patch_size, ln_eps=self.ln_eps, precision=self._precision,
                                         device=dev, pos_embed_mode=self.pos_embed_mode)
            self._engine_sig = sig
        return self._engine_obj

    def set_precision(self, precision) -> None:
        self._precision = parse_precision(precision)

    # ---- forward glue (dino.py:164-210 / ibot.py:182-220 / mocov3.py:146-186 / mae.py:195-237)
    def _tap_bn(self):
        if not self.add_norm:
            return None, 2
        bns = [dict(weight=bn.weight, bias=bn.bias, running_mean=bn.running_mean, running_var=bn.running_var,
                    num_batches_tracked=bn.num_batches_tracked) for bn in self.batchnorms]
        return bns, (0 if self.training else 1)

    def _extract(self, images: torch.Tensor, n_spatial_from_grid: bool = True, want_cls: bool = False):
        if not images.is_cuda:
            raise lib.MvpError("images must be on the HIP device (no CPU fallback)")
        eng = self.engine()
        bns, mode = self._tap_bn()
        with torch.no_grad():
            taps = eng.forward_taps(images, self.multilayers, bn=bns, bn_mode=mode, tap_input_of_block=self.tap_input_of_block,
                                    want_cls=want_cls or self.output in ("cls", "dense-cls"), groups=pipeline.current_groups())
            # (num_batches_tracked is incremented by the tap kernel's statistics pass: no extra launch)
        return taps

    def extract_kqv(self, images):
        """dino.py:82-139, ibot.py:128-180 (the same code in both wrappers): K / Q / V of the last attention layer (the output of its fused qkv projection), CLS row dropped,
        as [B, C, h*w] ("kqv": [B, 3C, h*w], k | q | v).  No centre padding on this path (the reference calls prepare_tokens directly;
        ``fixed_size`` is a multiple of the patch size in every config)."""
        if images.ndim == 3:
            images = images.unsqueeze(0)
        if images.ndim == 5:
            images = images.squeeze(0)
        if not images.is_cuda:
            raise lib.MvpError("images must be on the HIP device (no CPU fallback)")
        if images.shape[-2] % self.patch_size or images.shape[-1] % self.patch_size:
            raise ValueError("extract_kqv: the image size must be a multiple of the patch size (dino.py:105 has no padding on this path)")
        with torch.no_grad():
            qkv = self.engine().last_block_qkv(images)
        bs, _, C3 = qkv.shape
        C = C3 // 3
        hw = (images.shape[-2] // self.patch_size) * (images.shape[-1] // self.patch_size)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        pick = {"k": [k], "q": [q], "v": [v], "kqv": [k, q, v]}[self.mode_selected]
        return torch.cat([t[:, 1:].transpose(1, 2).reshape(bs, C, hw) for t in pick], dim=1)

    def preprocess_image(self, rgb_image):
        """dino.py:141-161, ibot.py:102-124: torchvision ``Resize((fixed_size, fixed_size))`` of a [C,H,W] / [B,C,H,W] tensor (bilinear, antialiased)."""
        from . import functional as MF

        x = rgb_image if rgb_image.ndim == 4 else rgb_image.unsqueeze(0)
        return MF.resize_antialias(x, (self.fixed_size, self.fixed_size)), self.fixed_size // self.patch_size, self.fixed_size // self.patch_size

    def _finish(self, taps: TapOutputs):
        """tokens_to_output (evals/models/utils.py:105-124) per tap.  'dense' is the kernel's own output; 'cls' is the
        (tap-normalised) CLS token the tap kernel emits next to the map; 'gap' / 'dense-cls' are shape glue on those.
        A grouped forward (TapGroups: one TapOutputs per batch) is finished batch by batch -> pipeline.GroupedFeatures."""
        if isinstance(taps, TapGroups):
            return pipeline.GroupedFeatures(self._finish(t) for t in taps)
        if self.output == "dense":
            outs = list(taps)
        elif self.output == "cls":
            outs = list(taps.cls)
        elif self.output == "gap":
            outs = [t.mean(dim=(2, 3)) for t in taps]
        elif self.output == "dense-cls":
            outs = [torch.cat((t, c[:, :, None, None].expand(-1, -1, t.shape[2], t.shape[3])), dim=1).contiguous() for t, c in zip(taps, taps.cls)]
        else:
            raise ValueError(f"unknown output type {self.output!r}")
        if self.output != "dense":
            return outs[0] if len(outs) == 1 else outs
        return taps[0] if len(taps) == 1 else taps
