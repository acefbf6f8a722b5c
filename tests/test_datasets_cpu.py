"""N3 host logic on CPU: batch-dict contract, build_loader's sampler semantics (builder.py:39-67)."""
import torch

from evals.datasets import SyntheticNYU, build_loader
from mvp import dist as mdist


def test_sample_contract_and_determinism():
    ds = SyntheticNYU("train", num_samples=6, image_size=(32, 48), max_depth=10.0)
    s0, s0b = ds[0], ds[0]
    assert s0["image"].shape == (3, 32, 48) and s0["depth"].shape == (1, 32, 48) and s0["snorm"].shape == (3, 32, 48)
    assert all(v.dtype == torch.float32 for k, v in s0.items() if k != "segmentation")
    seg = s0["segmentation"]  # nyu.py:245-251 carries the OneFormer panoptic map the per-segment metrics read
    assert seg.dtype == torch.int64 and seg.shape == (32, 48) and 0 <= int(seg.min()) and int(seg.max()) < 150
    assert all(torch.equal(s0[k], s0b[k]) for k in s0)  # pure function of the index
    assert not torch.equal(ds[1]["image"], s0["image"])
    assert not torch.equal(SyntheticNYU("valid", 6, (32, 48))[0]["image"], s0["image"])
    d = s0["depth"]
    assert (d == 0).float().mean().item() > 0.02 and d.max().item() <= 10.0 and d[d > 0].min().item() >= 0.05
    assert torch.allclose(s0["snorm"].norm(dim=0), torch.ones(32, 48), atol=1e-5)


def test_build_loader_single_process():
    ds = SyntheticNYU("train", num_samples=10, image_size=(16, 16))
    tr = build_loader(ds, "train", 4)
    va = build_loader(ds, "valid", 4)
    assert len(tr) == 3 and tr.sampler.__class__.__name__ == "RandomSampler" and va.sampler.__class__.__name__ == "SequentialSampler"
    sizes = [b["image"].shape[0] for b in va]
    assert sizes == [4, 4, 2]  # drop_last=False
    b = next(iter(va))
    assert set(b) == {"image", "depth", "snorm", "segmentation"} and b["depth"].shape == (4, 1, 16, 16)


def test_build_loader_distributed_sampler_matches_shard_indices():
    ds = SyntheticNYU("train", num_samples=11, image_size=(8, 8))
    seen = []
    for r in range(2):
        ld = build_loader(ds, "train", 4, num_gpus=2, rank=r)
        ld.sampler.set_epoch(3)
        idx = list(ld.sampler)
        # same permutation/padding/striding rule as mvp.dist.shard_indices (DistributedSampler semantics)
        assert idx == mdist.shard_indices(11, r, 2, epoch=3, shuffle=True, seed=0)
        seen += idx
    assert len(seen) == 12 and set(seen) == set(range(11))


def test_build_loader_from_config_node():
    ld = build_loader({"name": "synthetic", "image_size": [16, 24], "num_batches": 3, "batch_size": 2, "max_depth": 10}, "train", 2, with_snorm=False)
    b = next(iter(ld))
    assert len(ld) == 3 and set(b) == {"image", "depth", "segmentation"} and b["image"].shape == (2, 3, 16, 24)
