#!/usr/bin/env python3
"""Diagnostic: DevicePrefetcher feeding a 4-deep FeaturePipeline with its old buffer-recycling rule (consumer_lag 0) and with the lag the
pipeline asks for (3), against the serial loop: the old rule overwrites targets the pipeline still holds (trajectory differs), the new one
reproduces the serial loop bit for bit."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import test_gpu_pipeline as T
from evals.utils.losses import DepthLoss
from mvp import pipeline
from mvp.pipeline import FeaturePipeline
from mvp.prefetch import DevicePrefetcher
from mvp.train import train_depth_step
dev = torch.device("cuda:0")
host = []
for s in range(14):
    g = torch.Generator().manual_seed(700 + s)
    host.append({"image": torch.randn(4, 3, 64, 80, generator=g).pin_memory(), "depth": (torch.rand(4, 1, 64, 80, generator=g) * 9 + 0.05).pin_memory()})
def run(depth, lag):
    model, probe, opt, sched = T._build(dev)
    loss_fn = DepthLoss()
    pre = DevicePrefetcher(host, dev, depth=2, consumer_lag=lag)
    pipe = FeaturePipeline(model, depth)
    it = iter(pre); pend = []; losses = []
    def feed():
        try: b = next(it)
        except StopIteration: return False
        pipe.submit(b["image"]); pend.append(b); return True
    while True:
        while len(pipe) < pipe.depth and feed(): pass
        if not pend: break
        b = pend.pop(0)
        losses.append(train_depth_step(model, probe, opt, sched, loss_fn, None, b["depth"], feats=pipe.next()))
    torch.cuda.synchronize()
    return torch.stack(losses).cpu().numpy()
ref = run(1, 0)
ok = run(4, 3)
bad = run(4, 0)
print("lag 3 equals serial:", np.array_equal(ok, ref), "| lag 0 (the old recycling rule) equals serial:", np.array_equal(bad, ref))
