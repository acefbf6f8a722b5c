#!/usr/bin/env python3
"""Full-size sanity of the secondary BASELINE configs (finite losses, no faults, index ranges): not a parity test.
  #3 MoCo-v3 ResNet-50 multilayer + SurfaceNormalHead(dpt, UA) @480^2      (per-GPU batch 8)
  #4 MAE ViT-B/16 + DepthHead(dpt) on 512x512                               (per-GPU batch 8)
  #5 iBOT ViT-B/16 dense features at 800x800, one image pair + 20 keypoints"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch

dev = torch.device("cuda")
from evals.models.probes import DepthHead, SurfaceNormalHead
from evals.utils.losses import DepthLoss
from mvp.optim import FlatAdamW
from mvp.train import train_depth_step, train_snorm_step


def timed(f, n=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = f()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / n


g = torch.Generator().manual_seed(0)
which = sys.argv[1:] or ["3", "4", "5"]
if "4" in which:
    from evals.models.mae import MAE
    B = int(os.environ.get("B", 8))
    m = MAE(return_multilayer=True).to(dev)
    m.resize_pos_embed(image_size=(512, 512))  # train_depth.py:613-617
    probe = DepthHead(feat_dim=m.feat_dim, head_type="dpt", kernel_size=3, prediction_type="bindepth", hidden_dim=512).to(dev)
    opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}])
    img = torch.randn(B, 3, 512, 512, generator=g).to(dev)
    tgt = (torch.rand(B, 1, 512, 512, generator=g) * 9.9 + 0.05).to(dev)
    loss, dt = timed(lambda: train_depth_step(m, probe, opt, None, DepthLoss(), img, tgt.clone()))
    print(f"#4 MAE + DPT depth 512^2 B={B}: loss {loss.item():.4f}  {dt * 1e3:.1f} ms/step  {B / dt:.1f} img/s", flush=True)
    assert torch.isfinite(loss)
    del m, probe, opt
if "3" in which:
    from evals.models.mocov3_res50 import MoCoV3_RES
    B = int(os.environ.get("B", 8))
    m = MoCoV3_RES(return_layers=[1, 2, 3, 4], return_multilayer=True, add_norm=True).to(dev)  # configs/backbone/mocov3_resnet50.yaml
    probe = SurfaceNormalHead(feat_dim=m.feat_dim, head_type="dpt", uncertainty_aware=True, hidden_dim=512, kernel_size=3).to(dev)
    opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}])
    img = torch.randn(B, 3, 480, 480, generator=g).to(dev)
    n = torch.randn(B, 3, 480, 480, generator=g)
    tgt = (n / n.norm(dim=1, keepdim=True)).to(dev)
    mask = (torch.rand(B, 1, 480, 480, generator=g) > 0.1).to(dev)
    loss, dt = timed(lambda: train_snorm_step(m, probe, opt, None, img, tgt, mask))
    print(f"#3 MoCo-v3 R50 + DPT snorm(UA) 480^2 B={B}: loss {loss.item():.4f}  {dt * 1e3:.1f} ms/step  {B / dt:.1f} img/s", flush=True)
    assert torch.isfinite(loss)
    del m, probe, opt
if "5" in which:
    from evals.models.ibot import iBOT
    from mvp import spair
    m = iBOT(output="dense", layer=-1).to(dev)
    imgs = torch.randn(2, 3, 800, 800, generator=g).to(dev)
    kp = torch.rand(20, 2, generator=g)
    def pair():
        f = m(imgs)
        return spair.correspondence(f[0], f[1], kp)
    (xy, val), dt = timed(pair)
    print(f"#5 iBOT 800^2 pair + 20 keypoints: feats {tuple(m(imgs).shape)} argmax range x[{int(xy[:,0].min())},{int(xy[:,0].max())}] y[{int(xy[:,1].min())},{int(xy[:,1].max())}]  {dt * 1e3:.1f} ms/pair", flush=True)
    assert xy.min() >= 0 and xy.max() < 50 and torch.isfinite(val).all()
print("fullsize ok")
