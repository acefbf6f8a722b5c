#!/usr/bin/env python3
"""Diagnostic: split-K factor sweep on the hot-path GEMM shapes with too few output tiles (HIP events, interleaved)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from mvp import lib, ops

dev = torch.device("cuda")
B = int(os.environ.get("B", 16)); M = B * 197
shapes = [("proj", M, 768, 768), ("fc2", M, 768, 3072), ("head", B * 196, 256, 3072), ("qkv", M, 2304, 768), ("fc1", M, 3072, 768)]
for prec in (3, 1):
    for name, m, n, k in shapes:
        a = ops.split_bf16(torch.randn(m, k, device=dev), 3); w = ops.split_bf16(torch.randn(n, k, device=dev) * 0.05, 3)
        out = ops.empty_pair((m, n), 3, dev); bias = torch.randn(n, device=dev); res = torch.randn(m, n, device=dev); o32 = torch.empty(m, n, device=dev)
        res_t = {}
        for rnd in range(3):
            for S in (1, 2, 3, 4, 6, 8):
                if k // 64 < S:
                    continue
                f = lambda: ops.gemm(a, w, m, n, k, bias=bias, residual=res, out_f32=o32, out=out, precision=prec, splitk=S)
                for _ in range(3): f()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): f()
                e1.record(); torch.cuda.synchronize()
                res_t.setdefault(S, []).append(e0.elapsed_time(e1) / 20 * 1e3)
        fl = 2.0 * m * n * k
        print(f"prec={prec} {name:5s} M={m} N={n} K={k} auto={ops.splitk_auto(m, n, k)}: " + "  ".join(f"S{S}={min(v):6.1f}us({fl / min(v) / 1e6:4.0f})" for S, v in res_t.items()), flush=True)
