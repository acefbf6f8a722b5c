#!/usr/bin/env python3
"""Cheap bisect harness for the intermittent serial-vs-pipelined mismatch: ONE model / probe / optimiser, reset between
repetitions, REPS short trajectories through train_depth_step, compared bit for bit (losses, final parameters) with the serial
reference.  Variants (VAR=...): sync = host sync after every step; graphs = replayed forwards; d2 = two chains; nodefer = (unused)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import test_gpu_pipeline as T
from evals.utils.losses import DepthLoss
from evals.utils.optim import cosine_decay_linear_warmup
from mvp.pipeline import FeaturePipeline
from mvp.train import train_depth_step

VAR = set(os.environ.get("VAR", "").split(",")) - {""}
if "splitk" in VAR:  # round 1's automatic 4-way split-K of the probe head GEMM
    from mvp import ops as _ops
    _ops.splitk_auto = lambda M, N, K: 4 if (N <= 256 and K >= 2048 and ((M + 127) // 128) * ((N + 63) // 64) <= 128) else 1
dev = torch.device("cuda:0")
n = int(os.environ.get("STEPS", "7"))
B = int(os.environ.get("B", "4"))
hw = tuple(int(v) for v in os.environ.get("HW", "64x80").split("x"))
model, probe, opt, _ = T._build(dev)
loss_fn = DepthLoss()
bs = T._batches(dev, n, B=B, hw=hw)
init_p = opt.flat_param.clone()
init_bn = [(b.running_mean.clone(), b.running_var.clone(), b.num_batches_tracked.clone()) for b in model.batchnorms]


def reset():
    opt.flat_param.copy_(init_p)
    opt.exp_avg.zero_()
    opt.exp_avg_sq.zero_()
    opt._step = 0
    for b, (m, v, c) in zip(model.batchnorms, init_bn):
        b.running_mean.copy_(m); b.running_var.copy_(v); b.num_batches_tracked.copy_(c)
    for g in opt.param_groups:
        g["lr"] = 1e-3
        g.pop("initial_lr", None)
    return torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda e: cosine_decay_linear_warmup(e, 100, 10))


def run(pipe):
    sched = reset()
    losses, nxt = [], 0
    for i in range(n):
        while len(pipe) < pipe.depth and nxt < n:
            pipe.submit(bs[nxt]["image"])
            nxt += 1
        losses.append(train_depth_step(model, probe, opt, sched, loss_fn, None, bs[i]["depth"].clone(), feats=pipe.next()))
        if "sync" in VAR:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return torch.stack(losses).cpu(), opt.flat_param.clone(), model.batchnorms[3].running_var.clone()


serial = FeaturePipeline(model, 1)
ref = run(serial)
again = run(serial)
assert torch.equal(ref[0], again[0]) and torch.equal(ref[1], again[1]), "the serial loop does not reproduce itself"
depth = 2 if "d2" in VAR else int(os.environ.get("DEPTH", "3"))
pipe = FeaturePipeline(model, depth, graphs="graphs" in VAR)
reps = int(os.environ.get("REPS", "300"))
bad, first_steps = 0, {}
t0 = time.perf_counter()
for r in range(reps):
    got = run(pipe)
    if not (torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])):
        bad += 1
        ld = [i for i in range(n) if got[0][i] != ref[0][i]]
        key = ld[0] if ld else ("params-only" if not torch.equal(got[1], ref[1]) else "bn-only")
        first_steps[key] = first_steps.get(key, 0) + 1
print(f"VAR={sorted(VAR)} depth {depth} B={B} {hw}: {bad} of {reps} trajectories differ ({time.perf_counter() - t0:.1f} s); first differing loss step -> count: {first_steps}", flush=True)
