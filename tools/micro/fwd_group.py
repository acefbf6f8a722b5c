#!/usr/bin/env python3
"""A few grouped frozen forwards (G stacked batches of B images, 224^2, bf16x3) and nothing else: the profiling target for the
kernels of the grouped forward (rocprofv3 --kernel-trace --stats -- python3 tools/micro/fwd_group.py)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from evals.models.dino import DINO
from mvp import backbone as bb, pipeline
from mvp.train import extract_features

dev = torch.device("cuda:0")
B, G, n = int(os.environ.get("B", 16)), int(os.environ.get("G", 6)), int(os.environ.get("N", 6))
model = DINO(return_multilayer=True, add_norm=True, weights=bb.random_vit_state_dict(seed=0)).to(dev)
g = torch.Generator().manual_seed(0)
imgs = torch.randn(G * B, 3, 224, 224, generator=g).to(dev)
for _ in range(n):
    with pipeline._slot(0, 1, G):
        pipeline._take_deferred()
        extract_features(model, imgs)
        pipeline._take_deferred()
torch.cuda.synchronize()
print("done", G, B, n)
