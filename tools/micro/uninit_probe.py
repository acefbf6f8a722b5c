#!/usr/bin/env python3
"""Diagnostic: does any kernel of the step consume uninitialised memory?  torch.empty() is made to return NaN-filled tensors
(torch.utils.deterministic.fill_uninitialized_memory under use_deterministic_algorithms), then the small test trajectory runs serially
and pipelined: any NaN in features / loss / gradient / parameters points at a read of bytes nobody wrote."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
import test_gpu_pipeline as T
from evals.utils.losses import DepthLoss
from mvp.pipeline import FeaturePipeline
from mvp.train import train_depth_step

dev = torch.device("cuda:0")
print("empty() probe:", torch.empty(4, device=dev), flush=True)
for depth, graphs, kind in ((1, False, "linear"), (3, False, "linear"), (1, False, "dpt")):
    model, probe, opt, sched = T._build(dev, kind)
    loss_fn = DepthLoss()
    pipe = FeaturePipeline(model, depth, graphs=graphs)
    bs = T._batches(dev, 5)
    nxt = 0
    for i in range(5):
        while len(pipe) < pipe.depth and nxt < 5:
            pipe.submit(bs[nxt]["image"])
            nxt += 1
        f = pipe.next()
        nan_f = sum(int(torch.isnan(t).sum()) for t in f)
        loss = train_depth_step(model, probe, opt, sched, loss_fn, None, bs[i]["depth"].clone(), feats=f)
        torch.cuda.synchronize()
        print(f"{kind} depth {depth} step {i}: NaN in features {nan_f}, loss {float(loss):.6f}, NaN in flat grad {int(torch.isnan(opt.flat_grad).sum())}, "
              f"NaN in params {int(torch.isnan(opt.flat_param).sum())}", flush=True)
