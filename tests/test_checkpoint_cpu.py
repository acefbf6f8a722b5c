"""N2: checkpoint round trip in the reference's {"cfg","model","probe"} format, incl. loading a
reference-style DDP checkpoint ("module." prefixes) — host logic only, runs on CPU."""
import torch

from oracle import probes as oprobes


def _boom():
    FakeDictConfig.executed = True
    return FakeDictConfig()


class FakeDictConfig:  # stands in for omegaconf.DictConfig (not installed): unpickling it would run _boom()
    executed = False

    def __reduce__(self):
        return (_boom, ())


def test_probe_checkpoint_roundtrip(tmp_path):
    from evals.models.probes import DepthHead
    from mvp import checkpoint as ck

    probe = DepthHead(feat_dim=[32] * 4, head_type="dpt", prediction_type="bindepth", hidden_dim=16, kernel_size=3)
    sd = oprobes.make_dpt_weights([32] * 4, 256, hidden=16, k=3, seed=1)
    probe.load_state_dict(sd, strict=True)
    model = torch.nn.Linear(2, 2)
    path = ck.save_checkpoint(str(tmp_path / "exp" / "ckpt.pth"), {"note": "x"}, model, probe)
    blob = torch.load(path, weights_only=True)  # plain dicts + tensors only: nothing in the file needs executing
    assert set(blob) == {"cfg", "model", "probe"} and set(blob["probe"]) == set(sd)
    # reference checkpoints written from DDP-wrapped modules carry "module." prefixes
    blob["probe"] = {"module." + k: v for k, v in blob["probe"].items()}
    torch.save(blob, path)
    probe2 = DepthHead(feat_dim=[32] * 4, head_type="dpt", prediction_type="bindepth", hidden_dim=16, kernel_size=3)
    ck.load_checkpoint(path, model, probe2)
    for k, v in probe2.state_dict().items():
        assert torch.equal(v, sd[k]), k
    assert probe2.name == "bindepth_dpt_k3"


def test_checkpoint_with_object_cfg_is_refused_not_unpickled(tmp_path):
    """A reference ckpt.pth pickles cfg as an omegaconf DictConfig (train_depth.py:832-844).  load_checkpoint uses the
    weights_only loader: an arbitrary pickled object must be refused with a clear error, never executed; and a non-plain cfg
    handed to save_checkpoint is flattened so that our own files always load."""
    import pytest

    from evals.models.probes import DepthHead
    from mvp import checkpoint as ck

    probe = DepthHead(feat_dim=[32] * 4, head_type="linear", prediction_type="sigdepth", kernel_size=1)
    model = torch.nn.Linear(2, 2)
    bad = str(tmp_path / "ref_style.pth")
    torch.save({"cfg": FakeDictConfig(), "model": model.state_dict(), "probe": probe.state_dict()}, bad)
    with pytest.raises(RuntimeError, match="weights_only"):
        ck.load_checkpoint(bad, model, probe)
    assert FakeDictConfig.executed is False
    good = ck.save_checkpoint(str(tmp_path / "own.pth"), {"obj": FakeDictConfig(), "n": 3, "l": (1, 2)}, model, probe)
    out = ck.load_checkpoint(good, model, probe)
    assert out["cfg"]["n"] == 3 and out["cfg"]["l"] == [1, 2] and isinstance(out["cfg"]["obj"], str)


def test_ssl_resnet50_wrapper_surface():
    """N4: the other ResNet-50 SSL wrappers share the template: constructor surface + attributes (CPU, no compute)."""
    import importlib
    import warnings

    for mod, cls, name in (("barlowtwins", "BARLOWTWINS", "$barlowtwins$_resnet50_dense_[1, 2, 3, 4]"),
                           ("swav", "SWAV", "$swav$_resnet50_dense_[1, 2, 3, 4]"),
                           ("simclr", "SIMCLR", "simclr_resnet50_dense_[1, 2, 3, 4]"),
                           ("deepclusterv2", "DEEPCLUSTERV2", "$deepcluster_v2$_resnet50_dense_[1, 2, 3, 4]")):
        K = getattr(importlib.import_module(f"evals.models.{mod}"), cls)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = K(arch="resnet50", output="dense", return_layers=[1, 2, 3, 4], add_norm=True, return_multilayer=True)
        assert m.checkpoint_name == name and m.layer == "1-2-3-4" and m.patch_size == 0
        assert m.feat_dim == [(256, 120), (512, 60), (1024, 30), (2048, 15)] and len(m.batchnorms) == 5
        assert "model.layer4.2.conv3.weight" in m.state_dict()
