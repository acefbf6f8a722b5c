#!/usr/bin/env python3
"""Diagnostic: split factor sweep of the TN weight-gradient kernel on the DPT 3x3 conv shapes."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from mvp import lib, ops, conv

dev = torch.device("cuda")
for (B, H, C, Co) in ((16, 28, 512, 512), (8, 64, 512, 512), (16, 56, 256, 256), (8, 128, 512, 256), (16, 112, 256, 128)):
    M = B * H * H
    x = ops.split_bf16(torch.randn(M, C, device=dev), 3); g = ops.split_bf16(torch.randn(M, max(Co, 128), device=dev) * 1e-2, 3)
    geo = conv.geom(B, H, H, C, 3, 3, 1, 1)
    dw = torch.empty(Co, C, 3, 3, device=dev)
    res = {}
    for rnd in range(2):
        for S in (None, 2, 4, 7, 10, 14, 21, 32, 48, 64):
            f = lambda: conv.conv_dw(g, max(Co, 128), x, C, geo, Co, dw, precision=3, splits=S)
            for _ in range(2): f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(S, []).append(e0.elapsed_time(e1) / 5 * 1e3)
    fl = 2.0 * M * C * Co * 9
    print(f"B={B} {H}x{H} Cin={C} Cout={Co} M={M}: " + "  ".join(f"S{S}={min(v):7.1f}us({fl / min(v) / 1e6:4.0f})" for S, v in res.items()), flush=True)
