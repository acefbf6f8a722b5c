#!/usr/bin/env python3
"""Diagnostic: ResNet-50 (DINO wrapper, 4 taps, inputs resized to 480x480) frozen-forward throughput + kernel trace."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from evals.models.dino_res50 import DINO_RESNET

dev = torch.device("cuda")
B = int(os.environ.get("B", 16))
m = DINO_RESNET(return_multilayer=True, add_norm=True).to(dev)
x = torch.randn(B, 3, 224, 224, device=dev)
for _ in range(3):
    out = m(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    out = m(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"ResNet-50 multilayer extract B={B}: {dt * 1e3:.2f} ms/step, {B / dt:.1f} img/s, taps {[tuple(o.shape) for o in out]}")

# the same forwards kept in flight (mvp/pipeline.py: INFLIGHT batches ahead on up to 3 side streams, eager launches)
if os.environ.get("INFLIGHT", "4") != "1":
    from mvp.pipeline import FeaturePipeline

    for depth, streams, graphs in ((2, 2, False), (4, 3, False), (4, 3, True)):
        pipe = FeaturePipeline(m, depth, streams=streams, graphs=graphs)
        xs = [torch.randn(B, 3, 224, 224, device=dev) for _ in range(4)]

        def run(n):
            nxt = 0
            for i in range(n):
                while len(pipe) < pipe.depth and nxt < n:
                    pipe.submit(xs[nxt % 4])
                    nxt += 1
                pipe.next()

        run(8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(40)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 40
        print(f"ResNet-50 multilayer extract B={B}, {depth} batches ahead on {pipe.chains} streams, hipGraph replay {pipe.graphs}: {dt * 1e3:.2f} ms/step, {B / dt:.1f} img/s")

# ---- per-layer roofline of the trunk (serial forward): HIP events around every GEMM / implicit-GEMM convolution launch, grouped by shape.
# algorithmic flops = 2 M N K of the convolution; bytes = operands read once (bf16 pairs) + outputs written once.
if os.environ.get("LAYERS", "1") != "0":
    import collections
    from mvp import conv as cv, ops

    recs = []
    g0, c0 = ops.gemm, cv.conv_gemm

    def t_gemm(a, w, M, N, K, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g0(a, w, M, N, K, **kw); e1.record()
        outb = (4 if kw.get("out_f32") is not None else 0) + (4 if kw.get("out") is not None else 0) + (4 if (kw.get("residual") is not None or kw.get("residual_pair") is not None) else 0)
        recs.append((f"1x1  M={M:6d} K={K:4d} N={N:4d}" + (" +id" if kw.get("residual_pair") is not None else ""), 2.0 * M * N * K, M * K * 4 + N * K * 4 + M * N * outb, e0, e1))

    def t_conv(x, g, wk, N, **kw):
        M, K = g["B"] * g["Ho"] * g["Wo"], g["kh"] * g["kw"] * g["C"]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); c0(x, g, wk, N, **kw); e1.record()
        outb = (4 if kw.get("out_f32") is not None else 0) + (4 if kw.get("out") is not None else 0)
        recs.append((f"{g['kh']}x{g['kw']}/{g['stride']} M={M:6d} K={K:4d} N={N:4d}", 2.0 * M * N * K, g["B"] * g["H"] * g["W"] * g["C"] * 4 // (1 << (2 * g["up"])) + N * K * 4 + M * N * outb, e0, e1))

    ops.gemm, cv.conv_gemm = t_gemm, t_conv
    try:
        for _ in range(3):
            recs.clear()
            m(x)
            torch.cuda.synchronize()
    finally:
        ops.gemm, cv.conv_gemm = g0, c0
    agg = collections.OrderedDict()
    for name, fl, by, e0, e1 in recs:
        a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += fl; a[2] += by; a[3] += e0.elapsed_time(e1) * 1e-3
    tot = sum(a[3] for a in agg.values())
    print(f"per-layer (B={B}, one serial forward, {len(recs)} GEMM / conv launches, {tot * 1e3:.2f} ms of the {dt * 1e3:.2f} ms step): launches, us each, algorithmic TFLOP/s (frac of 2.5 PF), GB/s of compulsory traffic (frac of 8 TB/s)")
    for name, (n, fl, by, sec) in agg.items():
        print(f"  {name:34s} x{n:2d}  {sec / n * 1e6:7.1f} us  {fl / sec / 1e12:6.1f} TF/s ({fl / sec / 2.5e15:.3f})  {by / sec / 1e9:7.0f} GB/s ({by / sec / 8e12:.2f})  {100 * sec / tot:4.1f} %")
