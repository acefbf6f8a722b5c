"""NYU-shaped synthetic dataset honouring the reference's per-sample dict contract (evals/datasets/nyu.py:245-251):
{"image": float32 [3,H,W] (ImageNet-normalised statistics), "depth": float32 [1,H,W] metres with 0 = invalid,
 "snorm": float32 [3,H,W] unit vectors, "segmentation": int64 [H,W] OneFormer ADE20K ids in 16x16 blobs}.  Samples are a pure function of (seed, split, index), so every rank and every
epoch sees the same sample for the same index — what DistributedSampler sharding assumes."""
from __future__ import annotations

import torch
from torch.utils.data import Dataset


class SyntheticNYU(Dataset):
    def __init__(self, split: str = "train", num_samples: int = 128, image_size=(480, 640), max_depth: float = 10.0,
                 invalid_fraction: float = 0.1, seed: int = 0, with_snorm: bool = True, name: str = "synthetic"):
        self.split, self.n, self.hw = split, int(num_samples), tuple(int(v) for v in image_size)
        self.max_depth, self.invalid_fraction, self.seed, self.with_snorm, self.name = float(max_depth), float(invalid_fraction), int(seed), with_snorm, name
        self._salt = {"train": 0, "valid": 1, "val": 1, "test": 2}.get(split, 3)

    def __len__(self) -> int:
        return self.n

    def __getitem__(self, i: int):
        if not 0 <= i < self.n:
            raise IndexError(i)
        g = torch.Generator().manual_seed((self.seed * 4 + self._salt) * 1_000_003 + i)
        H, W = self.hw
        image = torch.randn(3, H, W, generator=g)
        depth = torch.rand(1, H, W, generator=g) * (self.max_depth - 0.1) + 0.05
        depth[torch.rand(1, H, W, generator=g) < self.invalid_fraction] = 0.0
        out = {"image": image, "depth": depth}
        if self.with_snorm:
            n = torch.randn(3, H, W, generator=g)
            out["snorm"] = n / n.norm(dim=0, keepdim=True).clamp_min(1e-6)
        blobs = torch.randint(0, 150, ((H + 15) // 16, (W + 15) // 16), generator=g)
        out["segmentation"] = blobs.repeat_interleave(16, 0).repeat_interleave(16, 1)[:H, :W].contiguous()
        return out
