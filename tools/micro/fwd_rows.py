#!/usr/bin/env python3
"""Frozen forward time against the number of images stacked into it (224^2, bf16x3): how the 256x256 tile grid's rounds on 256 CUs
quantise it.  T images -> M = 197 T rows -> ceil(M / 256) row tiles x {9, 3, 12, 3} column tiles (qkv, proj, fc1, fc2)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from evals.models.dino import DINO
from mvp import backbone as bb, pipeline
from mvp.train import extract_features

dev = torch.device("cuda:0")
model = DINO(return_multilayer=True, add_norm=True, weights=bb.random_vit_state_dict(seed=0)).to(dev)
g = torch.Generator().manual_seed(0)
for T in [int(a) for a in sys.argv[1:]] or [96, 104, 108, 110, 111, 112, 192, 220]:
    imgs = torch.randn(T, 3, 224, 224, generator=g).to(dev)
    ts = []
    for it in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with pipeline._slot(0, 1, 1):
            pipeline._take_deferred()
            e0.record()
            extract_features(model, imgs)
            e1.record()
            pipeline._take_deferred()
        torch.cuda.synchronize()
        if it >= 2:
            ts.append(e0.elapsed_time(e1))
    rt = -(-T * 197 // 256)
    print(f"T={T:4d} rows={T * 197:6d} row_tiles={rt:3d} (x3 = {rt * 3}): {min(ts):7.3f} ms  -> {T / min(ts) * 1e3:7.0f} img/s forward-only", flush=True)
