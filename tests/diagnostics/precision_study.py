#!/usr/bin/env python3
"""CPU emulation (diagnostic): how far do ViT-B/16 tap features drift from fp32 when every GEMM / attention
operand is rounded to a 16-bit format (fp32 accumulate)?  Decides which MFMA operand formats can meet the
1e-3 rel tolerance.  Uses the oracle as the fp32 model; nothing here is on the product path."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch, torch.nn.functional as F
from oracle import vit as ov

MODE = {"fmt": None, "a_split": False, "w_split": False, "comp": False}
S = 2.0 ** -6  # MVP_PREC_F16X2's share (include/mvp_hip.h)
_lin = F.linear


def rnd(x, fmt):
    return x.to(fmt).float()


def operand(x, split):
    fmt = MODE["fmt"]
    if fmt is None:
        return x
    hi = rnd(x, fmt)
    if split:
        return hi + rnd(x - hi, fmt)
    return hi


def f16(x):
    return x.clamp(-65504.0, 65504.0).half().float()


def linear(x, w, b=None):
    if MODE["comp"]:  # MVP_PREC_F16X2: a_hi . w_hi + a_lo . w_lo over the compensated fp16 pairs (the four GEMMs of a block)
        a_hi = f16(x)
        a_lo = f16((x - a_hi) * 8.0 + a_hi * 0.125)
        wd = w.double()
        w_hi = ((1.0 - S) * wd).half()
        w_lo = ((wd + ((1.0 - S) * wd - w_hi.double()) / S) / 8.0).half().float()
        y = _lin(a_hi, w_hi.float()) + _lin(a_lo, w_lo)
        return y if b is None else y + b
    return _lin(operand(x, MODE["a_split"]), operand(w, MODE["w_split"]), b)


def attention(sd, prefix, x, heads):
    B, N, C = x.shape
    d = C // heads
    qkv = linear(x, sd[prefix + "qkv.weight"], sd.get(prefix + "qkv.bias"))
    qkv = qkv.reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = (operand(t, MODE["a_split"]) for t in (qkv[0], qkv[1], qkv[2]))
    a = ((q @ k.transpose(-2, -1)) * (d ** -0.5)).softmax(dim=-1)
    y = (operand(a, MODE["a_split"]) @ v).transpose(1, 2).reshape(B, N, C)
    return linear(y, sd[prefix + "proj.weight"], sd[prefix + "proj.bias"])


def run(sd, img, real_scale):
    class _F:
        linear = staticmethod(linear)
        gelu = staticmethod(F.gelu)
        layer_norm = staticmethod(F.layer_norm)
        def __getattr__(self, k):
            return getattr(F, k)
    ov.F = _F()
    ov.attention = attention
    out = ov.vit_dense_features(sd, img, layers=[2, 5, 8, 11], heads=12, patch=16, add_norm=False, output="dense", return_tokens=True)
    return out


def main():
    torch.manual_seed(0)
    size = int(os.environ.get("SIZE", 224))
    sd = ov.make_vit_weights(seed=3)
    scale = float(os.environ.get("WSCALE", 1.0))
    if scale != 1.0:  # crude stand-in for trained checkpoints: larger weight magnitudes -> sharper attention, bigger residuals
        for k in sd:
            if k.endswith("weight") and sd[k].dim() == 2:
                sd[k] = sd[k] * scale
    img = torch.randn(2, 3, size, size)
    res = {}
    for name, fmt, a_s, w_s in (("fp32", None, 0, 0), ("bf16", torch.bfloat16, 0, 0), ("bf16x3~", torch.bfloat16, 1, 1),
                                ("fp16", torch.float16, 0, 0), ("fp16 a-split", torch.float16, 1, 0), ("fp16 w-split", torch.float16, 0, 1),
                                ("f16x2 (comp)", torch.bfloat16, 1, 1)):  # the last: compensated GEMMs, attention operands as bf16 pairs
        MODE.update(fmt=fmt, a_split=bool(a_s), w_split=bool(w_s), comp=name.startswith("f16x2"))
        with torch.no_grad():
            res[name] = run(sd, img, scale)
    ref = res["fp32"]
    ref = ref if isinstance(ref, (list, tuple)) else [ref]
    for name, o in res.items():
        o = o if isinstance(o, (list, tuple)) else [o]
        print(f"{name:14s}", "  ".join(f"{((a - b).norm() / b.norm()).item():.2e}" for a, b in zip(o, ref)), flush=True)


if __name__ == "__main__":
    main()
