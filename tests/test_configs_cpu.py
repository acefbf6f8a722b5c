"""The hydra config surface (north_star: "keeps the hydra backbone/probe config surface"):
  * this package's configs/<group>/<choice>.yaml compose through mvp.config exactly like `python train_depth.py backbone=dino_b16`;
  * (build container only) every hot-path choice file of the REFERENCE loads and its ``_target_`` + kwargs instantiate against this
    package unchanged — i.e. the reference's own yaml drives these classes."""
import os

import pytest
import yaml

REF = "/root/reference/configs"
BACKBONES = ["dino_b16", "ibot_b16", "mae_b16", "mocov3_b14", "dino_resnet50", "mocov3_resnet50"]
PROBES = ["depth_dpt", "snorm_dpt"]


def test_compose_per_choice_layout():
    from mvp import config

    cfg = config.compose("depth_training", ["backbone=ibot_b16", "probe=depth_linear", "+backbone.return_multilayer=True", "batch_size=4",
                                            "optimizer.probe_lr=0.001"])
    assert cfg["backbone"]["_target_"] == "evals.models.ibot.iBOT" and cfg["backbone"]["return_multilayer"] is True
    assert cfg["probe"]["head_type"] == "linear" and cfg["batch_size"] == 4 and cfg["optimizer"]["probe_lr"] == 0.001
    assert cfg["system"]["random_seed"] == 8 and cfg["optimizer"]["n_epochs"] == 10
    for group in ("backbone", "probe", "optimizer", "dataset"):
        assert os.path.isdir(os.path.join(config.CONFIG_DIR, group)), f"configs/{group}/ must hold one file per choice (hydra layout)"
        assert not os.path.exists(os.path.join(config.CONFIG_DIR, group + ".yaml"))
    with pytest.raises(KeyError):
        config.compose("depth_training", ["backbone=does_not_exist"])


def test_own_choice_files_match_reference_schema():
    """Same keys and values as the reference's files for the hot-path choices (a config schema is category-(b) similarity)."""
    if not os.path.isdir(REF):
        pytest.skip("reference configs only exist in the build container")
    from mvp import config

    for group, names in (("backbone", BACKBONES), ("probe", PROBES), ("optimizer", ["one_epoch", "three_epoch", "ten_epoch", "fifteen_epoch"])):
        for n in names:
            ref = yaml.safe_load(open(os.path.join(REF, group, n + ".yaml")))
            own = yaml.safe_load(open(os.path.join(config.CONFIG_DIR, group, n + ".yaml")))
            assert own == ref, (group, n, own, ref)


@pytest.mark.parametrize("name", BACKBONES)
def test_reference_backbone_yaml_instantiates_here(name):
    if not os.path.isdir(REF):
        pytest.skip("reference configs only exist in the build container")
    import warnings

    from mvp import config

    node = yaml.safe_load(open(os.path.join(REF, "backbone", name + ".yaml")))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # "no local checkpoint: seeded random init"
        model = config.instantiate(node, return_multilayer=True)  # train_depth.py:564-567 adds return_multilayer for dpt probes
    assert type(model).__module__ == node["_target_"].rsplit(".", 1)[0]
    assert len(model.feat_dim) == 4 and model.output == node["output"] and hasattr(model, "checkpoint_name") and hasattr(model, "patch_size")


@pytest.mark.parametrize("name", PROBES)
def test_reference_probe_yaml_instantiates_here(name):
    if not os.path.isdir(REF):
        pytest.skip("reference configs only exist in the build container")
    from mvp import config

    node = yaml.safe_load(open(os.path.join(REF, "probe", name + ".yaml")))
    probe = config.instantiate(node, feat_dim=[768] * 4)
    assert probe.name == {"depth_dpt": "bindepth_dpt_k3", "snorm_dpt": "snorm_dpt_k3_UA"}[name]
    n = sum(p.numel() for p in probe.parameters())
    if name == "depth_dpt":
        assert n == 38_151_936  # SURVEY §8 P4 (verified against the reference module)
