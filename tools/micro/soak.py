#!/usr/bin/env python3
"""Soak of the default training loop shape: several "epochs" of pipelined_features (a fresh pipeline per epoch, as mvp.train.train does),
a validation pass in eval mode between them; device memory after each epoch must plateau (slot buffers, carry stores and graphs of
finished pipelines are released), losses stay finite, throughput steady."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from evals.models.dino import DINO
from evals.models.probes import DepthHead
from evals.utils.losses import DepthLoss
from mvp import backbone as bb, pipeline
from mvp.optim import FlatAdamW
from mvp.pipeline import pipelined_features
from mvp.train import train_depth_step

dev = torch.device("cuda:0")
B, n = 16, int(os.environ.get("N", 120))
model = DINO(return_multilayer=True, add_norm=True, weights=bb.random_vit_state_dict(seed=0)).to(dev)
probe = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type="bindepth", min_depth=0.001, max_depth=10).to(dev)
opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}])
g = torch.Generator().manual_seed(0)
batches = [(torch.randn(B, 3, 224, 224, generator=g).to(dev), (torch.rand(B, 1, 224, 224, generator=g) * 9.9 + 0.05).to(dev)) for _ in range(8)]
pipeline.freeze_gc()
mem = []
for ep in range(int(os.environ.get("EPOCHS", 6))):
    model.train()
    t0 = time.perf_counter()
    last = None
    for (img, tgt), f in pipelined_features(model, [batches[i % 8] for i in range(n)], probe=probe):
        last = train_depth_step(model, probe, opt, None, DepthLoss(), None, tgt.clone(), feats=f)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    model.eval()
    with torch.no_grad():
        nval = sum(1 for _ in pipelined_features(model, [batches[i % 8] for i in range(23)], probe=probe))
    torch.cuda.synchronize()
    import gc; gc.collect()
    mem.append(torch.cuda.memory_allocated() / 2**20)
    print(f"epoch {ep}: {n * B / dt:8.0f} img/s  loss {last.item():.4f}  validated {nval} batches  allocated {mem[-1]:9.1f} MiB  reserved {torch.cuda.memory_reserved() / 2**20:9.1f} MiB", flush=True)
    assert torch.isfinite(last)
assert mem[-1] <= mem[2] * 1.02 + 64, ("device memory keeps growing", mem)
print("soak ok")
