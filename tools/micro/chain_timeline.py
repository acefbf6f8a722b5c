#!/usr/bin/env python3
"""Diagnostic: device timeline of the pipelined headline step (eager launches, HIP events around every GEMM / attention / LayerNorm /
tap-BN launch on its own stream).  For D forwards in flight prints ms/step, the average duration of each kernel kind while chains
overlap, how many traced kernels are in flight on average, and which kinds run beside the GEMMs.  Why is D=4 worse than D=3?"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from evals.models.dino import DINO
from evals.models.probes import DepthHead
from evals.utils.losses import DepthLoss
from mvp import backbone as bb, ops
from mvp.optim import FlatAdamW
from mvp.pipeline import FeaturePipeline, freeze_gc
from mvp.train import train_depth_step

dev = torch.device("cuda")
B = 16
model = DINO(return_multilayer=True, add_norm=True, weights=bb.random_vit_state_dict(seed=0)).to(dev)
probe = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type="bindepth", min_depth=0.001, max_depth=10).to(dev)
opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}])
loss_fn = DepthLoss()
batches = [(torch.randn(B, 3, 224, 224, device=dev), torch.rand(B, 1, 224, 224, device=dev) * 9 + 0.05) for _ in range(4)]
freeze_gc()


def run(pipe, n):
    nxt = 0
    for i in range(n):
        while len(pipe) < pipe.depth and nxt < n:
            pipe.submit(batches[nxt % 4][0])
            nxt += 1
        train_depth_step(model, probe, opt, None, loss_fn, None, batches[i % 4][1], feats=pipe.next())


for depth in [int(x) for x in os.environ.get("DEPTHS", "3,4,5,6").split(",")]:
    pipe = FeaturePipeline(model, depth, graphs=False)
    run(pipe, 12)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(pipe, 40)
    torch.cuda.synchronize()
    untraced = (time.perf_counter() - t0) / 40 * 1e3
    trace = []
    base = torch.cuda.Event(enable_timing=True)
    base.record()
    ops.set_trace(trace)
    t0 = time.perf_counter()
    run(pipe, 24)
    torch.cuda.synchronize()
    traced = (time.perf_counter() - t0) / 24 * 1e3
    ops.set_trace(None)
    ev = []
    for kind, tile, prec, work, e0, e1 in trace:
        name = {"gemm": "gemm", "attention": "attn"}.get(kind, tile.split()[0][:9])
        ev.append((base.elapsed_time(e0), base.elapsed_time(e1), name))
    ev.sort()
    lo, hi = ev[len(ev) // 4][0], ev[3 * len(ev) // 4][0]  # steady middle half
    mid = [e for e in ev if lo <= e[0] <= hi]
    dur = {}
    for a, b, n in mid:
        d = dur.setdefault(n, [0, 0.0])
        d[0] += 1
        d[1] += b - a
    # average number of traced kernels in flight, and what runs beside a GEMM
    busy = sum(b - a for a, b, n in mid)
    span = max(b for a, b, n in mid) - min(a for a, b, n in mid)
    beside = {}
    gem = [(a, b) for a, b, n in mid if n == "gemm"]
    for a, b, n in mid:
        ov = 0.0
        for ga, gb in gem:
            if gb <= a:
                continue
            if ga >= b:
                break
            if (ga, gb) != (a, b):
                ov += max(0.0, min(b, gb) - max(a, ga))
        e = beside.setdefault(n, [0.0, 0.0])
        e[0] += ov
        e[1] += b - a
    print(f"D={depth}: {untraced:.3f} ms/step untraced, {traced:.3f} traced | in flight (traced kernels) avg {busy / span:.2f} | " +
          " ".join(f"{n} {1e3 * t / c:.1f}us" for n, (c, t) in sorted(dur.items())) + " | GEMM-overlap share: " +
          " ".join(f"{n} {o / max(t, 1e-9):.2f}" for n, (o, t) in sorted(beside.items())), flush=True)
