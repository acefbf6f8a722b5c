"""The fourteen ResNet-50 SSL wrappers of the reference that differ only in a checkpoint-key prefix, a checkpoint file name and a
name tag (evals/models/{barlowtwins,byol,clusterfit,deepclusterv2,densecl,jigsaw,mocov2,npid,pirl,rotnet,selav2,simclr,simsiam,
swav}.py: template clones of mocov3_res50.py).  One table instead of fourteen files; ``evals.models.<name>`` stays importable
(hydra ``_target_: evals.models.byol.BYOL``) because each entry is registered as a sub-module of this package."""
from __future__ import annotations

import sys
import types

from mvp.resnet_backbone import make_ssl_resnet50

# module name -> (class name, checkpoint_name tag, state-dict prefixes to strip, local checkpoint file stems)
SSL_RESNET50 = {
    "barlowtwins": ("BARLOWTWINS", "$barlowtwins$", ["backbone."], ["barlowtwins_resnet50"]),
    "byol": ("BYOL", "$byol$", ["module."], ["byol_resnet50"]),
    "clusterfit": ("CLUSTERFIT", "$clusterfit$", ["_feature_blocks."], ["clusterfit_resnet50"]),
    "deepclusterv2": ("DEEPCLUSTERV2", "$deepcluster_v2$", ["module."], ["deepclusterv2_resnet50"]),
    "densecl": ("DENSECL", "$densecl$", [], ["densecl_resnet50"]),
    "jigsaw": ("JIGSAW", "$jigsaw$", ["_feature_blocks."], ["jigsaw_resnet50"]),
    "mocov2": ("MOCOV2", "$mocov2$", ["module.encoder_q."], ["mocov2_resnet50", "moco_v2_800ep_pretrain"]),
    "npid": ("NPID", "$npid$", ["_feature_blocks."], ["npid_resnet50"]),
    "pirl": ("PIRL", "$pirl$", ["_feature_blocks."], ["pirl_resnet50"]),
    "rotnet": ("ROTNET", "$rotnet$", ["_feature_blocks."], ["rotnet_resnet50"]),
    "selav2": ("SELAV2", "$sela_v2$", ["module."], ["selav2_resnet50"]),
    "simclr": ("SIMCLR", "simclr", ["_feature_blocks."], ["simclr_resnet50"]),
    "simsiam": ("SIMSIAM", "$simsiam$", ["backbone."], ["simsiam_resnet50"]),
    "swav": ("SWAV", "$swav$", ["module."], ["swav_resnet50", "swav_800ep_pretrain"]),
}


def register(package: str = "evals.models") -> None:
    """Create ``<package>.<name>`` modules holding the wrapper class, so ``import evals.models.byol`` and hydra targets resolve."""
    pkg = sys.modules[package]
    for mod_name, (cls_name, tag, prefixes, files) in SSL_RESNET50.items():
        full = f"{package}.{mod_name}"
        if full in sys.modules:
            continue
        cls = make_ssl_resnet50(cls_name, tag, prefixes, files, f"evals/models/{mod_name}.py")
        m = types.ModuleType(full, f"{full}.{cls_name} — drop-in for the reference's evals/models/{mod_name}.py (shared ResNet-50 SSL template)")
        setattr(m, cls_name, cls)
        cls.__module__ = full
        sys.modules[full] = m
        setattr(pkg, mod_name, m)
