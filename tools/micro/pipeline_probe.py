#!/usr/bin/env python3
"""Diagnostic for mvp/pipeline.py: per-step host enqueue time vs device time of the headline step (B=16, 224^2, linear probe) for
1 / 2 / 3 forwards in flight and several run lengths -- is a pipelined run host-bound, clock-bound or neither?"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from evals.models.dino import DINO
from evals.models.probes import DepthHead
from evals.utils.losses import DepthLoss
from mvp import backbone as bb
from mvp.optim import FlatAdamW
from mvp.pipeline import FeaturePipeline
from mvp.train import train_depth_step

dev = torch.device("cuda")
B = int(os.environ.get("B", "16"))
model = DINO(return_multilayer=True, add_norm=True, weights=bb.random_vit_state_dict(seed=0)).to(dev)
probe = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type="bindepth", min_depth=0.001, max_depth=10).to(dev)
opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}])
loss_fn = DepthLoss()
batches = [(torch.randn(B, 3, 224, 224, device=dev), torch.rand(B, 1, 224, 224, device=dev) * 9 + 0.05) for _ in range(4)]


def run(pipe, n):
    nxt = 0
    for i in range(n):
        while len(pipe) < pipe.depth and nxt < n:
            pipe.submit(batches[nxt % 4][0])
            nxt += 1
        train_depth_step(model, probe, opt, None, loss_fn, None, batches[i % 4][1], feats=pipe.next())


for depth in (1, 2, 3):
    pipe = FeaturePipeline(model, depth, run_ahead=int(os.environ.get("AHEAD", "8")))
    run(pipe, 10)
    torch.cuda.synchronize()
    for n in (30, 100, 300, 30):
        time.sleep(float(os.environ.get("REST", "0.5")))
        t0 = time.perf_counter()
        run(pipe, n)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"inflight {depth} steps {n:4d}: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, total {1e3 * (t2 - t0) / n:.3f} ms/step = {B * n / (t2 - t0):.0f} img/s", flush=True)
