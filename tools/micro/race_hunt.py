#!/usr/bin/env python3
"""Diagnostic for an intermittent 1e-6 loss mismatch between the serial and the pipelined (3 streams, eager launches) trajectory:
repeats the small test trajectory REPS times and reports, per step, whether the FEATURES or only the LOSS / weights differ from the
serial reference.  HEAD_SPLITK=4 brings back the probe head's automatic 4-way split-K GEMM, TILES=alone keeps the serial tile rule in the pipeline."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import test_gpu_pipeline as T
from evals.utils.losses import DepthLoss
from mvp import ops, pipeline
from mvp.pipeline import FeaturePipeline
from mvp.train import train_depth_step

if os.environ.get("HEAD_SPLITK") == "4":  # round 1's automatic rule (the default is no split-K since the race was found)
    ops.splitk_auto = lambda M, N, K: 4 if (N <= 256 and K >= 2048 and ((M + 127) // 128) * ((N + 63) // 64) <= 128) else 1
if os.environ.get("TILES") == "alone":
    pipeline.SHARED_TILES_FROM = 99
dev = torch.device("cuda:0")
n = 7


from mvp import functional as MF


def run(depth, graphs=False):
    """One trajectory through train_depth_step itself; a forward hook keeps the probe's output, and the flat gradient / parameters are
    copied after every step (3 MB each: light enough not to hide the race, which the fully opened-up body did)."""
    model, probe, opt, sched = T._build(dev)
    loss_fn = DepthLoss()
    pipe = FeaturePipeline(model, depth, graphs=graphs)
    bs = T._batches(dev, n)
    log, nxt, outs = [], 0, []
    h = probe.register_forward_hook(lambda m, i, o: outs.append(o.detach().clone()))
    for i in range(n):
        while len(pipe) < pipe.depth and nxt < n:
            pipe.submit(bs[nxt]["image"])
            nxt += 1
        f = pipe.next()
        loss = train_depth_step(model, probe, opt, sched, loss_fn, None, bs[i]["depth"].clone(), feats=f)
        log.append({"probe_out": outs.pop(), "loss": loss, "grad": opt.flat_grad.clone(), "param": opt.flat_param.clone()})
    h.remove()
    torch.cuda.synchronize()
    return log


ref = run(1)
bad = 0
reps = int(os.environ.get("REPS", "15"))
for r in range(reps):
    got = run(int(os.environ.get("DEPTH", "3")), graphs=os.environ.get("GRAPHS") == "1")
    first = None
    for i in range(n):
        for key in ("probe_out", "loss", "grad", "param"):
            if not torch.equal(got[i][key], ref[i][key]):
                d = (got[i][key].double() - ref[i][key].double()).abs()
                first = (i, key, int((d > 0).sum()), got[i][key].numel(), float(d.max()))
                break
        if first:
            break
    if first:
        bad += 1
        extra = ""
        if first[1] == "grad":
            d = (got[first[0]]["grad"] - ref[first[0]]["grad"]).abs()
            idx = torch.nonzero(d > 0).flatten()
            extra = f" | differing grad indices {int(idx.min())}..{int(idx.max())} of {d.numel()} (weight is [256, 3072] first, then bias)"
        print(f"rep {r}: first difference at step {first[0]} in '{first[1]}': {first[2]} of {first[3]} elements, max abs {first[4]:.3e}{extra}", flush=True)
print(f"{bad} of {reps} repetitions differ (HEAD_SPLITK={os.environ.get('HEAD_SPLITK', 'off')} TILES={os.environ.get('TILES', 'auto')} DEPTH={os.environ.get('DEPTH', '3')})", flush=True)
