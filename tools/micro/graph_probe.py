#!/usr/bin/env python3
"""Diagnostic: how much of the frozen ViT-B/16 forward's wall time is launch overhead?  Times model(images) eagerly and as a
captured hipGraph replay (torch.cuda.CUDAGraph captures the ctypes launches on the capturing stream)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from evals.models.dino import DINO
from mvp import backbone as bb

dev = torch.device("cuda")
model = DINO(return_multilayer=True, add_norm=True, weights=bb.random_vit_state_dict(seed=0)).to(dev)
x = torch.randn(16, 3, 224, 224, device=dev)


def timed(f, n=200):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    eager = timed(lambda: model(x))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            model(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = model(x)
    graphed = timed(g.replay)
print(f"ViT-B/16 4-tap forward B=16: eager {eager:.3f} ms, hipGraph replay {graphed:.3f} ms ({100 * (eager - graphed) / eager:.1f} % saved)")
