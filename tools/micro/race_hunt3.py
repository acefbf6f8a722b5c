#!/usr/bin/env python3
"""Which kernel of the probe step gives a different result when frozen forwards run beside it?  A side stream is kept busy with
eager forwards (the serial tile rule: the regime with the highest mismatch rate); on the main stream the stages of the probe step run
REPS times on FIXED inputs and every stage's output is compared with its first value."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import test_gpu_pipeline as T
from evals.utils.losses import DepthLoss
from mvp import functional as MF
from mvp.train import extract_features

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "4"))
hw = tuple(int(v) for v in os.environ.get("HW", "64x80").split("x"))
model, probe, opt, _ = T._build(dev)
loss_fn = DepthLoss()
bs = T._batches(dev, 3, B=B, hw=hw)
feats = [t.clone() for t in extract_features(model, bs[0]["image"])]
target = bs[0]["depth"]
side = torch.cuda.Stream()
LOAD = os.environ.get("LOAD", "1") == "1"


def stages():
    out = {}
    opt.zero_grad()
    p0 = probe(feats)
    out["probe_out"] = p0.detach().clone()
    p1 = MF.interpolate(p0, size=target.shape[-2:], mode="bilinear")
    out["pred"] = p1.detach().clone()
    loss = loss_fn(p1, target.clone())
    out["loss"] = loss.detach().clone()
    MF.backward(loss)
    opt._gather_stray_grads()
    out["grad"] = opt.flat_grad.clone()
    return out


torch.cuda.synchronize()
ref = stages()
torch.cuda.synchronize()
bad = {k: 0 for k in ref}
reps = int(os.environ.get("REPS", "300"))
for r in range(reps):
    if LOAD:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            model(bs[1 + r % 2]["image"])  # eager forward beside the probe stages (train-mode BN: running stats move, nobody reads them)
    got = stages()
    for k in ref:
        if not torch.equal(got[k], ref[k]):
            bad[k] += 1
torch.cuda.synchronize()
print(f"load={LOAD} B={B} {hw}: mismatches in {reps} repetitions per stage (cumulative along the chain): {bad}", flush=True)
