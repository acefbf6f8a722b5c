#!/usr/bin/env python3
"""Diagnostic: does running the frozen ViT-B/16 forward as TWO independent half-batch chains on two HIP streams (inside one captured
hipGraph, so both chains are really in flight together) fill the CUs better than one B=16 chain?  Each chain has its own model
instance (the per-model workspaces must not be shared between concurrent forwards); the weights are the same values at different
addresses, which overstates the L2 cost of a real implementation a little."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from evals.models.dino import DINO
from mvp import backbone as bb

dev = torch.device("cuda")
B = int(os.environ.get("B", "16"))
models = [DINO(return_multilayer=True, add_norm=True, weights=bb.random_vit_state_dict(seed=0)).to(dev) for _ in range(2)]
x = torch.randn(B, 3, 224, 224, device=dev)
halves = [x[: B // 2].contiguous(), x[B // 2:].contiguous()]


def timed(f, n=200):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def capture(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


side = torch.cuda.Stream()


def one_chain():
    return models[0](x)


def two_serial():
    return models[0](halves[0]), models[1](halves[1])


def two_parallel():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    a = models[0](halves[0])
    with torch.cuda.stream(side):
        b = models[1](halves[1])
    cur.wait_stream(side)
    return a, b


with torch.no_grad():
    res = {}
    for name, fn in (("one chain B", one_chain), ("two half-batch chains, serial", two_serial), ("two half-batch chains, two streams", two_parallel)):
        g = capture(fn)
        res[name] = timed(g.replay)
        print(f"{name:40s} {res[name]:.3f} ms  ({B / res[name] * 1e3:.0f} img/s forward only)", flush=True)
