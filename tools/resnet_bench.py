#!/usr/bin/env python3
"""Diagnostic: ResNet-50 (DINO wrapper, 4 taps, inputs resized to 480x480) frozen-forward throughput + kernel trace."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from evals.models.dino_res50 import DINO_RESNET

dev = torch.device("cuda")
B = int(os.environ.get("B", 16))
m = DINO_RESNET(return_multilayer=True, add_norm=True).to(dev)
x = torch.randn(B, 3, 224, 224, device=dev)
for _ in range(3):
    out = m(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    out = m(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"ResNet-50 multilayer extract B={B}: {dt * 1e3:.2f} ms/step, {B / dt:.1f} img/s, taps {[tuple(o.shape) for o in out]}")

# the same forwards kept in flight (mvp/pipeline.py: INFLIGHT batches ahead on up to 3 side streams, eager launches)
if os.environ.get("INFLIGHT", "4") != "1":
    from mvp.pipeline import FeaturePipeline

    for depth, streams, graphs in ((2, 2, False), (4, 3, False), (4, 3, True)):
        pipe = FeaturePipeline(m, depth, streams=streams, graphs=graphs)
        xs = [torch.randn(B, 3, 224, 224, device=dev) for _ in range(4)]

        def run(n):
            nxt = 0
            for i in range(n):
                while len(pipe) < pipe.depth and nxt < n:
                    pipe.submit(xs[nxt % 4])
                    nxt += 1
                pipe.next()

        run(8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(40)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 40
        print(f"ResNet-50 multilayer extract B={B}, {depth} batches ahead on {pipe.chains} streams, hipGraph replay {pipe.graphs}: {dt * 1e3:.2f} ms/step, {B / dt:.1f} img/s")
