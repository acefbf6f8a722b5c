#!/usr/bin/env python3
"""Diagnostic: throughput of mvp.train.train() — the reference-shaped loop with its per-step ``loss.item()`` host sync
(train_depth.py:143) — for the headline configuration, with the default pipeline and with MVP_INFLIGHT=1."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from evals.models.dino import DINO
from evals.models.probes import DepthHead
from evals.utils.losses import DepthLoss
from mvp import backbone as bb
from mvp.optim import FlatAdamW
from mvp.train import train

dev = torch.device("cuda")
B, n = int(os.environ.get("B", "16")), 200
model = DINO(return_multilayer=True, add_norm=True, weights=bb.random_vit_state_dict(seed=0)).to(dev)
probe = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type="bindepth", min_depth=0.001, max_depth=10).to(dev)
opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}])
src = [{"image": torch.randn(B, 3, 224, 224, device=dev), "depth": torch.rand(B, 1, 224, 224, device=dev) * 9 + 0.05} for _ in range(4)]


class Loader(list):
    sampler = None


for inflight in ("default", "1"):
    if inflight == "1":
        os.environ["MVP_INFLIGHT"] = "1"
    loader = Loader(src[i % 4] for i in range(n))
    train(model, probe, Loader(src), opt, None, 1, True, DepthLoss())  # warm-up epoch (4 batches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hist = train(model, probe, loader, opt, None, 1, True, DepthLoss())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"train() with loss.item() every step, MVP_INFLIGHT={inflight}: {B * n / dt:.0f} img/s ({1e3 * dt / n:.3f} ms/step), mean loss {hist[0]:.4f}", flush=True)
