"""build_loader with the reference's semantics (evals/datasets/builder.py:39-67): DistributedSampler iff num_gpus > 1,
shuffle only for the single-process train split, drop_last=False, pin_memory=True.  The reference hard-codes
num_workers=0 (builder.py:53-54); here it is a keyword (default 0 = identical behaviour) because once the step takes
3 ms the decode + collate of the next batch must overlap it (mvp.prefetch.DevicePrefetcher does the H2D side)."""
from __future__ import annotations

import torch
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler


def _make_dataset(cfg, split, **kwargs):
    if isinstance(cfg, Dataset):
        return cfg
    from mvp import config

    node = dict(cfg)
    if "task" in node:
        raise NotImplementedError("TaskonomyDataset wrapping (builder.py:47-49) is outside the hot path")
    if "_target_" not in node:  # configs/dataset/synthetic.yaml style
        from .synthetic import SyntheticNYU

        return SyntheticNYU(split=split, num_samples=node.get("num_samples", node.get("num_batches", 8) * node.get("batch_size", 16)),
                            image_size=node.get("image_size", (480, 640)), max_depth=node.get("max_depth", 10.0), **kwargs)
    return config.instantiate(node, split=split, **kwargs)


def build_loader(cfg, split, batch_size, num_gpus=1, num_workers=0, rank=None, **kwargs):
    dataset = _make_dataset(cfg, split, **kwargs)
    use_ddp = num_gpus > 1
    if use_ddp:
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            sampler = DistributedSampler(dataset)
        else:  # explicit rank / world (tests, pair sharding without a process group)
            sampler = DistributedSampler(dataset, num_replicas=num_gpus, rank=int(rank or 0))
    else:
        sampler = None
    shuffle = (split == "train") and not use_ddp
    return DataLoader(dataset, batch_size, num_workers=num_workers, drop_last=False, pin_memory=torch.cuda.is_available(),
                      shuffle=shuffle, sampler=sampler, persistent_workers=num_workers > 0)
