#!/usr/bin/env python3
"""Diagnostic: linear-head weight gradient dW[K, Ctot] = gl0^T . tok, NT GEMM on transposed packs vs the TN split-K kernel."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from mvp import lib, ops, conv

dev = torch.device("cuda")
for B in (16, 64):
    h = w = 14; M = B * h * w; K4 = 256; C = 3072; Mpad = (M + 63) // 64 * 64
    tok32 = torch.randn(M, C, device=dev); g32 = torch.randn(M, K4, device=dev) * 1e-3
    tok = ops.split_bf16(tok32, 3)
    tokT = ops.zeros_pair((C, Mpad), 3, dev); gT = ops.zeros_pair((K4, Mpad), 3, dev)
    for pair, src in ((tokT, tok32), (gT, g32)):
        t = ops.split_bf16(src.t().contiguous(), 3)
        pair[0][:, :M].copy_(t[0]); pair[1][:, :M].copy_(t[1])
    ref = g32.double().t() @ tok32.double()
    dW0 = torch.empty(K4, C, device=dev); dW1 = torch.empty(K4, C, device=dev)
    geo = dict(B=B, H=h, W=w, C=C, Ho=h, Wo=w, kh=1, kw=1, stride=1, pad=0, up=0)

    def nt():
        ops.pack_nchw_tokens(g32, 1, M, K4, tok=gT, ld_tok=Mpad, col_off=0)
        ops.gemm(gT, tokT, K4, C, Mpad, out_f32=dW0, precision=3)

    def tn(splits):
        gp = ops.split_bf16(g32, 3)
        conv.conv_dw(gp, K4, tok, C, geo, K4, dW1, precision=3, splits=splits)

    def timeit(f, *a):
        for _ in range(3): f(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f(*a)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3

    print(f"B={B} NT(pack+gemm) {timeit(nt):.1f}us  err {(dW0.double()-ref).norm()/ref.norm():.2e}", flush=True)
    for s in (1, 2, 4, 6, 8, 12, 16):
        t = timeit(tn, s)
        print(f"   TN splits={s}: {t:.1f}us  err {(dW1.double()-ref).norm()/ref.norm():.2e}", flush=True)
