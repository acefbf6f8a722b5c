#!/usr/bin/env python3
"""Diagnostic: the pipelined evaluate_dataset rate of config #5 with the pipeline's resolved shape (depth / group / span / graphs) printed."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from evals.models.ibot import iBOT
from mvp import backbone as bb, pipeline, spair

dev = torch.device("cuda")
model = iBOT(return_multilayer=False, add_norm=True, weights=bb.random_vit_state_dict(seed=0)).to(dev)
ds = [spair.SyntheticSPair(num_pairs=32, image_size=800, num_kps=20)[i] for i in range(32)]
spair.evaluate_dataset(model, ds[:6], 0.1)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    recall, _ = spair.evaluate_dataset(model, ds, 0.1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{32 / dt:.1f} pairs/s ({1e3 * dt / 32:.2f} ms/pair)", {k: (p.depth, p.group, p.span, p.graphs, p.chains, len(p._graphs)) for k, p in pipeline.cached_pipelines(model).items()},
          "qk16", model.engine().att_qk_f16, flush=True)
