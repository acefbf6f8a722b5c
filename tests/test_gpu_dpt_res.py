"""GPU parity of the DPT probe's ResNet-pyramid variant (probes.py:312-350 + pre-activation
ResidualConvUnits + bilinear x2 fusion; BASELINE config #3's head) against the CPU oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from test_gpu_dpt import _MaskedRelu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


DIMS = [(128, 0), (128, 0), (256, 0), (128, 0)]


def _pyramid(B, s, g):
    return [torch.randn(B, DIMS[i][0], s[0] * 2 ** (3 - i), s[1] * 2 ** (3 - i), generator=g) for i in range(4)]


def _emulate(sd, feats, ctx, B):
    """fp64 restatement of the CNN-variant DPT with the HIP path's ReLU gates pinned."""
    from mvp import dpt_res

    P = {n: t.clone().double().requires_grad_(True) for n, t in sd.items()}
    cl = lambda m, H, W: m.cpu().reshape(B, H, W, -1).permute(0, 3, 1, 2).double()
    r = []
    for i in range(4):
        H, W = feats[i].shape[-2:]
        r.append(_MaskedRelu.apply(F.conv2d(feats[i].double(), P[f"head.conv_{i}.weight"], None, padding=1), cl(ctx.mf[i], H, W)))

    def rcu(x, pre, n, extra=None, relu_out=False):
        _, _, ma, my, H, W = ctx.saved[n]
        a = _MaskedRelu.apply(F.conv2d(x, P[pre + "conv1.weight"], P[pre + "conv1.bias"], padding=1), cl(ma, H, W))
        t = F.conv2d(a, P[pre + "conv2.weight"], P[pre + "conv2.bias"], padding=1) + x
        if extra is not None:
            t = t + extra
        return _MaskedRelu.apply(t, cl(my, H, W)) if relu_out else t

    up = None
    for n, (blk, unit) in enumerate(dpt_res.RCU_ORDER):
        pre = f"head.ref_{blk}.resConfUnit{unit}."
        if unit == 1:
            cur = rcu(r[blk], pre, n, extra=up, relu_out=True)
        else:
            y = rcu(r[3] if blk == 3 else cur, pre, n)
            up = F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=True)
    H2, W2 = up.shape[-2:]
    h0 = _MaskedRelu.apply(F.conv2d(up, P["head.out_conv.0.weight"], P["head.out_conv.0.bias"], padding=1), cl(ctx.m0, H2, W2))
    return F.conv2d(h0, P["head.out_conv.2.weight"], P["head.out_conv.2.bias"], padding=1), P


@pytest.mark.parametrize("odim", [3, 256])
def test_dpt_res_backward_chain_with_pinned_gates(dev, odim):
    from evals.models.probes import DPT
    from mvp import dpt_res, ops
    from oracle import probes as oprobes

    B, Hd = 2, 128
    g = torch.Generator().manual_seed(30 + odim)
    feats = _pyramid(B, (3, 4), g)
    head = DPT([tuple(d) for d in DIMS], odim, Hd, 3)
    sd = oprobes.make_dpt_weights(DIMS, odim, hidden=Hd, k=3, seed=6)
    head.load_state_dict({k[len("head."):]: v for k, v in sd.items()}, strict=True)
    head = head.to(dev)
    toks, dims = [], []
    for f in feats:
        _, C, H, W = f.shape
        tok = ops.empty_pair((B * H * W, C), head.precision, dev)
        ops.pack_nchw_tokens(f.to(dev), B, C, H * W, tok=tok, ld_tok=C, col_off=0)
        toks.append(tok); dims.append((C, H, W))
    lq = dpt_res.dpt_res_logits(toks, dims, B, head, head.precision)
    gy = torch.randn(B, lq.shape[1], lq.shape[2], odim, generator=g)
    gyp = torch.zeros(lq.shape)
    gyp[..., :odim] = gy
    (lq * gyp.to(dev)).sum().backward()
    torch.cuda.synchronize()
    y, P = _emulate(sd, feats, lq.grad_fn, B)
    (y * gy.permute(0, 3, 1, 2).double()).sum().backward()
    assert rel_l2(lq.detach().cpu()[..., :odim].permute(0, 3, 1, 2).numpy(), y.detach().numpy()) < 5e-5
    for n, p in head.named_parameters():
        assert rel_l2(p.grad.cpu().numpy(), P["head." + n].grad.numpy()) < 1e-4, n


def test_dpt_res_heads_vs_oracle(dev):
    """SurfaceNormalHead(dpt) and DepthHead(dpt, sigdepth) on a ResNet pyramid vs the plain oracle."""
    from evals.models.probes import DepthHead, SurfaceNormalHead
    from oracle import probes as oprobes

    B, Hd = 2, 128
    g = torch.Generator().manual_seed(12)
    feats = _pyramid(B, (3, 4), g)
    fd = [tuple(d) for d in DIMS]
    for kind in ("snorm", "sigdepth"):
        if kind == "snorm":
            probe = SurfaceNormalHead(feat_dim=fd, head_type="dpt", uncertainty_aware=False, hidden_dim=Hd, kernel_size=3)
            odim = 3
        else:
            probe = DepthHead(feat_dim=fd, head_type="dpt", prediction_type="sigdepth", hidden_dim=Hd, kernel_size=3)
            odim = 1
        sd = oprobes.make_dpt_weights(DIMS, odim, hidden=Hd, k=3, seed=6)
        probe.load_state_dict(sd, strict=True)
        probe = probe.to(dev)
        y = probe([f.to(dev) for f in feats])
        sd_r = {n: t.clone().requires_grad_(True) for n, t in sd.items()}
        y_ref = oprobes.snorm_head(sd_r, [f.clone() for f in feats], "dpt", 3) if kind == "snorm" else \
            oprobes.depth_head(sd_r, [f.clone() for f in feats], "dpt", 3, "sigdepth")
        assert tuple(y.shape) == tuple(y_ref.shape)
        gy = torch.randn(y_ref.shape, generator=g)
        (y_ref * gy).sum().backward()
        (y * gy.to(dev)).sum().backward()
        torch.cuda.synchronize()
        assert rel_l2(y.detach().cpu().numpy(), y_ref.detach().numpy()) < 1e-4, kind
        for n, p in probe.named_parameters():
            a64, b64 = p.grad.double().cpu().flatten(), sd_r[n].grad.double().flatten()
            assert float((a64 - b64).norm() / b64.norm()) < 5e-2, (kind, n)       # ReLU-gate flips, see test_gpu_dpt.py
            assert 1 - float(a64 @ b64 / (a64.norm() * b64.norm())) < 2e-3, (kind, n)
