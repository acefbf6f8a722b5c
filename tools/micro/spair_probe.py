#!/usr/bin/env python3
"""Diagnostic: evaluate_dataset (config #5: iBOT ViT-B/16, 800x800 image pairs, 20 keypoints) pairs/s with the pairs' forwards in
flight and as one serial chain (MVP_INFLIGHT=1)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from evals.models.ibot import iBOT
from mvp import backbone as bb, spair

dev = torch.device("cuda")
model = iBOT(return_multilayer=False, add_norm=True, weights=bb.random_vit_state_dict(seed=0)).to(dev)


class Cached(list):  # dataset items prepared once: the timing is the device path, not the synthetic generator
    pass


ds = Cached(spair.SyntheticSPair(num_pairs=32, image_size=800, num_kps=20)[i] for i in range(32))
for inflight in ("default", "1"):
    if inflight == "1":
        os.environ["MVP_INFLIGHT"] = "1"
    spair.evaluate_dataset(model, ds[:6], 0.1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    recall, _ = spair.evaluate_dataset(model, ds, 0.1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"evaluate_dataset 32 pairs @800^2, MVP_INFLIGHT={inflight}: {32 / dt:.1f} pairs/s ({1e3 * dt / 32:.2f} ms/pair), recall {recall:.2f}", flush=True)
