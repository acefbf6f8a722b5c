#!/usr/bin/env python3
"""Diagnostic: per-step host-enqueue vs device-completion timeline of the headline step, alternating pipelined / serial legs.
For every leg: ms/step, the share of steps whose enqueue finished less than 0.3 ms before the device finished the previous step
(host-bound steps), and the distribution of device step-to-step intervals.  Tells a host-starved leg from a slow device."""
import os, sys, time, gc
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
B = 16
model = DINO(return_multilayer=True, add_norm=True, weights=bb.random_vit_state_dict(seed=0)).to(dev)
probe = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type="bindepth", min_depth=0.001, max_depth=10).to(dev)
opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}])
loss_fn = DepthLoss()
batches = [(torch.randn(B, 3, 224, 224, device=dev), torch.rand(B, 1, 224, 224, device=dev) * 9 + 0.05) for _ in range(4)]
AHEAD = int(os.environ.get("AHEAD", "0"))
if os.environ.get("NOGC"):
    gc.disable()


def run(pipe, n):
    e0 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    t0 = time.perf_counter()
    host, evs = [], []
    nxt = 0
    for i in range(n):
        while len(pipe) < pipe.depth and nxt < n:
            pipe.submit(batches[nxt % 4][0])
            nxt += 1
        train_depth_step(model, probe, opt, None, loss_fn, None, batches[i % 4][1], feats=pipe.next())
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        evs.append(e)
        host.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) * 1e3
    gpu = [e0.elapsed_time(e) for e in evs]
    return total, host, gpu


pipes = {1: FeaturePipeline(model, 1, run_ahead=AHEAD), 2: FeaturePipeline(model, 2, run_ahead=AHEAD)}
run(pipes[2], 10)
plan = [(2, 30), (1, 30), (2, 100), (1, 100), (2, 300), (1, 300)] * int(os.environ.get("REPS", "4"))
for depth, n in plan:
    total, host, gpu = run(pipes[depth], n)
    d = sorted(b - a for a, b in zip(gpu, gpu[1:]))
    starved = sum(1 for i in range(1, n) if host[i] > gpu[i - 1] - 0.3)
    hd = sorted(b - a for a, b in zip(host, host[1:]))
    print(f"inflight {depth} x{n:3d}: {total / n:.3f} ms/step {B * n / total * 1e3:6.0f} img/s | host-bound steps {starved:3d}/{n} | device interval p10/p50/p90/max "
          f"{d[len(d) // 10]:.2f}/{d[len(d) // 2]:.2f}/{d[len(d) * 9 // 10]:.2f}/{d[-1]:.2f} | host interval p50/p90/max {hd[len(hd) // 2]:.2f}/{hd[len(hd) * 9 // 10]:.2f}/{hd[-1]:.2f}", flush=True)
