#!/usr/bin/env python3
"""Does the 64x64 GEMM write global memory outside its output?  Operands and output are carved out of one arena with 0x5A guard bands
around the output (and around the operands); after REPS launches at ragged M the guards must be untouched."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from mvp import ops
from mvp.lib import PREC_BF16X3

dev = torch.device("cuda:0")
M, N, K = 84, 768, 3072
arena = torch.full((64 << 20,), 0x5A, dtype=torch.uint8, device=dev)
off = 1 << 20


def carve(nbytes, dtype, shape):
    global off
    t = arena[off:off + nbytes].view(dtype).view(shape)
    off += nbytes + (1 << 20)  # 1 MiB guard after every tensor
    return t


g = torch.Generator().manual_seed(1)
a32 = torch.randn(M, K, generator=g).to(dev)
w32 = (torch.randn(N, K, generator=g) * 0.05).to(dev)
a_hi = carve(M * K * 2, torch.bfloat16, (M, K)); a_lo = carve(M * K * 2, torch.bfloat16, (M, K))
w_hi = carve(N * K * 2, torch.bfloat16, (N, K)); w_lo = carve(N * K * 2, torch.bfloat16, (N, K))
h, l = ops.split_bf16(a32, PREC_BF16X3); a_hi.copy_(h); a_lo.copy_(l)
h, l = ops.split_bf16(w32, PREC_BF16X3); w_hi.copy_(h); w_lo.copy_(l)
out = carve(M * N * 4, torch.float32, (M, N))
o_hi = carve(M * N * 2, torch.bfloat16, (M, N)); o_lo = carve(M * N * 2, torch.bfloat16, (M, N))
used = [(t.data_ptr() - arena.data_ptr(), t.numel() * t.element_size()) for t in (a_hi, a_lo, w_hi, w_lo, out, o_hi, o_lo)]
mask = torch.ones(arena.numel(), dtype=torch.bool, device=dev)
for o, n in used:
    mask[o:o + n] = False
torch.cuda.synchronize()
for r in range(int(os.environ.get("REPS", "200"))):
    ops.gemm((a_hi, a_lo), (w_hi, w_lo), M, N, K, out_f32=out, out=(o_hi, o_lo))
torch.cuda.synchronize()
dirty = torch.nonzero((arena != 0x5A) & mask).flatten()
print(f"guard bytes modified: {dirty.numel()}" + (f" first at arena offset {int(dirty[0])}, last {int(dirty[-1])}; tensors at {used}" if dirty.numel() else ""), flush=True)
