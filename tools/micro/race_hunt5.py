#!/usr/bin/env python3
"""Reproducibility under concurrent load of the remaining kernels of the linear bin head's forward: the weight split, the NCHW ->
token-major packer, the fused bilinear x4 + bin expectation, and the resize / loss kernels behind it."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import test_gpu_pipeline as T
from evals.utils.losses import DepthLoss
from mvp import functional as MF, lib, ops
from mvp.lib import PREC_BF16X3
from mvp.train import extract_features

dev = torch.device("cuda:0")
model, probe, opt, _ = T._build(dev)
bs = T._batches(dev, 3, B=4, hw=(64, 80))
feats = [t.clone() for t in extract_features(model, bs[0]["image"])]
side = torch.cuda.Stream()
reps = int(os.environ.get("REPS", "400"))
w = torch.randn(256, 3072, device=dev) * 0.05
B, h, wd, K = 4, 4, 5, 256
l0 = torch.randn(B * h * wd, K, device=dev)
loss_fn = DepthLoss()
pred_fixed = torch.rand(4, 1, 64, 80, device=dev) * 9 + 0.1
pred_small = torch.rand(4, 1, 16, 20, device=dev) * 9 + 0.1


def t_split():
    hi, lo = ops.split_bf16(w, PREC_BF16X3)
    return [hi.clone(), lo.clone()]


def t_pack():
    p = MF.pack_features([f.clone() for f in feats], PREC_BF16X3)
    return [p.tok[0].clone(), p.tok[1].clone()]


def t_bins():
    P = B * 16 * h * wd
    depth = torch.empty(B, 1, 4 * h, 4 * wd, dtype=torch.float32, device=dev)
    inv = torch.empty(P, dtype=torch.float32, device=dev)
    gate = torch.empty(P, K // 8, dtype=torch.uint8, device=dev)
    lib.call("mvp_linear_bins_fwd", lib.LinearBinsArgs(lib.ptr(l0), lib.ptr(depth), lib.ptr(inv), lib.ptr(gate), None, None, B, h, wd, K, 4, 0.001, 10.0))
    return [depth, inv, gate]


def t_resize():
    return [MF.interpolate(pred_small, size=(64, 80), mode="bilinear")]


def t_loss():
    p = pred_fixed.clone().requires_grad_(True)
    loss = loss_fn(p, bs[0]["depth"].clone())
    MF.backward(loss)
    return [loss.detach().clone(), p.grad.clone()]


for name, fn in (("split_bf16", t_split), ("pack_nchw_tokens", t_pack), ("linear_bins_fwd", t_bins), ("resize_fwd", t_resize), ("depth_loss fwd+bwd", t_loss)):
    torch.cuda.synchronize()
    ref = fn()
    torch.cuda.synchronize()
    bad = 0
    for r in range(reps):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            model(bs[1 + r % 2]["image"])
        got = fn()
        if not all(torch.equal(a, b) for a, b in zip(got, ref)):
            bad += 1
    torch.cuda.synchronize()
    print(f"{name}: {bad} of {reps} differ under load", flush=True)
