#!/usr/bin/env python3
"""Is a GEMM launch's result reproducible while other kernels run beside it?  Fixed operands, REPS launches per shape on the main
stream, eager frozen forwards on a side stream as the load.  Shapes: the probe head (64x64 two-stage tile), a backbone projection
(64x64 single-stage), qkv (128x128)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import test_gpu_pipeline as T
from mvp import ops
from mvp.lib import PREC_BF16X3

dev = torch.device("cuda:0")
model, probe, opt, _ = T._build(dev)
bs = T._batches(dev, 3, B=int(os.environ.get("B", "4")), hw=(64, 80))
side = torch.cuda.Stream()
LOAD = os.environ.get("LOAD", "1")
reps = int(os.environ.get("REPS", "400"))
g = torch.Generator().manual_seed(3)
for (M, N, K) in ((80, 256, 3072), (80, 768, 768), (84, 2304, 768), (84, 768, 3072), (3152, 256, 3072), (3152, 768, 768)):
    a32 = torch.randn(max(M, 128), K, generator=g).to(dev)
    w32 = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    a, w = ops.split_bf16(a32, PREC_BF16X3), ops.split_bf16(w32, PREC_BF16X3)
    bias = torch.randn(N, generator=g).to(dev)
    out = torch.empty(M, N, dtype=torch.float32, device=dev)
    ops.gemm(a, w, M, N, K, bias=bias, out_f32=out)
    torch.cuda.synchronize()
    ref = out.clone()
    bad, nbad_el = 0, 0
    for r in range(reps):
        if LOAD == "1":
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model(bs[1 + r % 2]["image"])
        out.fill_(float("nan"))
        ops.gemm(a, w, M, N, K, bias=bias, out_f32=out)
        if not torch.equal(out, ref):
            bad += 1
            d = (out != ref) | torch.isnan(out)
            nbad_el = int(d.sum())
            rows = torch.nonzero(d.any(dim=1)).flatten()
            cols = torch.nonzero(d.any(dim=0)).flatten()
            where = f"rows {int(rows.min())}..{int(rows.max())} cols {int(cols.min())}..{int(cols.max())}, max abs {float((out - ref).abs().nan_to_num(1e30).max()):.3e}"
    torch.cuda.synchronize()
    print(f"load={LOAD} gemm M={M} N={N} K={K} tile {ops.gemm_tile(M, N, K)}: {bad} of {reps} launches differ" + (f" (last: {nbad_el} elements, {where})" if bad else ""), flush=True)
