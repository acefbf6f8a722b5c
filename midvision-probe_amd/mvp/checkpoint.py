"""End-of-run checkpoint in the reference's format (train_depth.py:526-539,832-844,
train_snorm.py:541-542): ``torch.save({"cfg", "model", "probe"}, exp_path / "ckpt.pth")``.

Probe state dicts are key-compatible with the reference in both directions (head.conv.*,
head.conv_i.*, head.ref_i.resConfUnit{1,2}.conv.{0,2}.* / conv{1,2}.*, head.out_conv.{0,2}.*), so a
reference ``ckpt.pth["probe"]`` loads here and vice versa.  Backbone state dicts keep the
reference layout for DINO / iBOT (``vit.*``, ``batchnorms.*``), MoCo-v3 (``model.*``) and the
ResNets (``model.*``); the MAE wrapper stores fused DINO-style keys under ``vit.*`` instead of
the HF ViT-MAE names (converted on load by mvp.backbone.hf_vitmae_to_fused)."""
from __future__ import annotations

import os
from typing import Any, Dict

import torch


def remove_module_prefix(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """train_depth.py:538-539 (DDP-wrapped reference checkpoints carry a ``module.`` prefix)."""
    return {key.replace("module.", ""): val for key, val in state_dict.items()}


def _plain(obj):
    """cfg as plain containers (what the weights_only loader accepts)."""
    if isinstance(obj, dict):
        return {str(k): _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    if isinstance(obj, (str, int, float, bool)) or obj is None:
        return obj
    return str(obj)


def save_checkpoint(path: str, cfg: Any, model: torch.nn.Module, probe: torch.nn.Module) -> str:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    # clone to contiguous CPU tensors: probe parameters are views into FlatAdamW's flat buffer
    ckpt = {"cfg": _plain(cfg),
            "model": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
            "probe": {k: v.detach().cpu().clone() for k, v in probe.state_dict().items()}}
    torch.save(ckpt, path)
    return path


def load_checkpoint(path: str, model: torch.nn.Module, probe: torch.nn.Module, load_model: bool = False) -> dict:
    """Depth eval loads the probe only (train_depth.py:533-535); snorm eval loads both (train_snorm.py:541-542)."""
    # weights_only=True: nothing from the file is executed.  save_checkpoint writes plain dicts + tensors, which this accepts.
    # A reference-written ckpt.pth pickles cfg as an omegaconf DictConfig (train_depth.py:832-844): the safe loader refuses
    # that object — convert such a file once where omegaconf is installed (torch.save({"probe": ckpt["probe"], "model": ...})).
    try:
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:  # pickle.UnpicklingError and friends
        raise RuntimeError(f"{path}: not loadable with weights_only=True ({type(e).__name__}: {str(e)[:200]}). Reference checkpoints carry "
                           "an omegaconf cfg object; re-save their 'probe' / 'model' state dicts as plain tensors first.") from e
    probe_sd = remove_module_prefix(ckpt["probe"])
    with torch.no_grad():  # copy in place: parameters may be views of the optimiser's flat buffer
        own = probe.state_dict()
        missing = set(own) ^ set(probe_sd)
        if missing:
            raise KeyError(f"probe state-dict keys differ: {sorted(missing)[:6]}")
        for k, v in probe_sd.items():
            own[k].copy_(v)
    if load_model:
        model.load_state_dict(remove_module_prefix(ckpt["model"]))
    return ckpt
