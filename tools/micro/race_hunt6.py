#!/usr/bin/env python3
"""linear_bins_fwd under load, one kind of load kernel at a time: which co-resident kernel disturbs it?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from mvp import lib, ops
from mvp.lib import PREC_BF16X3

dev = torch.device("cuda:0")
side = torch.cuda.Stream()
reps = int(os.environ.get("REPS", "400"))
B, h, wd, K = 4, 4, 5, 256
l0 = torch.randn(B * h * wd, K, device=dev)
g = torch.Generator().manual_seed(1)
M, C = 84, 768
x = torch.randn(M, C, generator=g).to(dev)
gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
xn = ops.empty_pair((M, C), PREC_BF16X3, dev)
qkv = ops.split_bf16(torch.randn(M, 3 * C, generator=g).to(dev), PREC_BF16X3)
ao = ops.empty_pair((M, C), PREC_BF16X3, dev)
a_p = ops.split_bf16(torch.randn(128, 3072, generator=g).to(dev), PREC_BF16X3)
w_p = ops.split_bf16((torch.randn(768, 3072, generator=g) * 0.05).to(dev), PREC_BF16X3)
w_q = ops.split_bf16((torch.randn(2304, 768, generator=g) * 0.05).to(dev), PREC_BF16X3)
out768 = torch.empty(M, 768, device=dev)
out128 = torch.empty(128, 768, device=dev)
outq = ops.empty_pair((M, 2304), PREC_BF16X3, dev)
bn_ws = torch.empty(ops.bn_tokens_workspace_bytes(M, C) // 4 + 16, dtype=torch.float32, device=dev)
stats = torch.empty(3 * C, device=dev)
nchw = torch.empty(4, C, 4, 5, device=dev)
img = torch.randn(4, 3, 64, 80, device=dev)
patches = ops.empty_pair((80, 768), PREC_BF16X3, dev)

loads = {
    "none": lambda: None,
    "gemm 64x64 (fc2 shape)": lambda: [ops.gemm(a_p, w_p, M, 768, 3072, out_f32=out768) for _ in range(6)],
    "gemm 64x64 M=128 (no ragged rows)": lambda: [ops.gemm(a_p, w_p, 128, 768, 3072, out_f32=out128) for _ in range(6)],
    "gemm 64x64 M=64 K=768": lambda: [ops.gemm(a_p, w_p, 64, 768, 768, out_f32=out128, lda=3072, ldw=3072) for _ in range(12)],
    "gemm 128x128 (qkv shape)": lambda: [ops.gemm(xn, w_q, M, 2304, 768, out=outq) for _ in range(6)],
    "attention": lambda: [ops.attention(qkv, ao, 4, 21, 12, 0.125, PREC_BF16X3) for _ in range(12)],
    "layernorm": lambda: [ops.layernorm(x, gam, bet, xn, M, C, 1e-6) for _ in range(24)],
    "bn_tokens": lambda: [ops.bn_tokens_to_nchw(x, 4, 21, C, 20, workspace=bn_ws, stats=stats, nchw=nchw, mode=0, defer_running=True) for _ in range(8)],
    "patch_gather": lambda: [ops.patch_gather(img, patches, 16, 4, 5, 0, 0) for _ in range(12)],
}


def bins():
    P = B * 16 * h * wd
    depth = torch.empty(B, 1, 4 * h, 4 * wd, dtype=torch.float32, device=dev)
    inv = torch.empty(P, dtype=torch.float32, device=dev)
    gate = torch.empty(P, K // 8, dtype=torch.uint8, device=dev)
    lib.call("mvp_linear_bins_fwd", lib.LinearBinsArgs(lib.ptr(l0), lib.ptr(depth), lib.ptr(inv), lib.ptr(gate), None, None, B, h, wd, K, 4, 0.001, 10.0))
    return [depth, inv, gate]


ops.layernorm(x, gam, bet, xn, M, C, 1e-6)
torch.cuda.synchronize()
ref = bins()
torch.cuda.synchronize()
only = os.environ.get("ONLY")
for name, load in loads.items():
    if only and only not in name:
        continue
    bad, which = 0, [0, 0, 0]
    for r in range(reps):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            load()
        got = bins()
        eq = [torch.equal(a, b) for a, b in zip(got, ref)]
        if not all(eq):
            bad += 1
            for i, e in enumerate(eq):
                which[i] += (not e)
            if bad <= 3:
                d = got[0].flatten() != ref[0].flatten()
                idx = torch.nonzero(d).flatten()
                gi = torch.nonzero((got[2] != ref[2]).any(dim=1)).flatten()
                same = torch.nonzero(ref[0].flatten() == got[0].flatten()[idx[0]]).flatten().tolist()
                Wo = 4 * wd
                print(f"   pixel {int(idx[0])} = (b {int(idx[0]) // (16 * h * wd)}, y {(int(idx[0]) // Wo) % (4 * h)}, x {int(idx[0]) % Wo}); reference pixels holding the value it got: {same}")
                print(f"   rep {r}: depth differs at {idx.numel()} pixels {idx[:12].tolist()} got {got[0].flatten()[idx[:6]].tolist()} want {ref[0].flatten()[idx[:6]].tolist()}; "
                      f"inv got {got[1][idx[:4]].tolist()} want {ref[1][idx[:4]].tolist()}; gate rows differing {gi[:12].tolist()}", flush=True)
    torch.cuda.synchronize()
    print(f"load '{name}': {bad} of {reps} differ (depth / inv_sum / gate mismatches: {which})", flush=True)
