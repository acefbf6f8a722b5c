"""GPU parity of the conv path and the DPT probe (reference default head, configs/probe/depth_dpt.yaml)
against the CPU oracle (itself pinned to the reference's DPT by tests/golden/probes.npz)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _cl_pair(x_nchw, dev):
    """NCHW fp32 cpu -> channels-last bf16 pair on device."""
    from mvp import ops
    return ops.split_bf16(x_nchw.permute(0, 2, 3, 1).contiguous().to(dev))


@pytest.mark.parametrize("cfg", [
    dict(B=2, H=12, W=10, Cin=64, Cout=96, k=3, up=0),
    dict(B=1, H=16, W=16, Cin=128, Cout=128, k=3, up=2),   # virtual nearest x4 input (out_conv.0)
    dict(B=2, H=9, W=7, Cin=32, Cout=4, k=1, up=0),
    dict(B=3, H=8, W=8, Cin=128, Cout=260, k=3, up=0),
])
def test_conv_forward_epilogues(dev, cfg):
    from mvp import conv as cv, lib, ops

    g = torch.Generator().manual_seed(cfg["Cin"] + cfg["Cout"])
    B, H, W, Cin, Cout, k, up = (cfg[n] for n in ("B", "H", "W", "Cin", "Cout", "k", "up"))
    xs = torch.randn(B, Cin, H >> up, W >> up, generator=g)
    x = F.interpolate(xs, scale_factor=2 ** up) if up else xs
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=k // 2).relu()
    ref_mask = (ref > 0)
    ref = ref + res.double()
    ge = cv.geom(B, H, W, Cin, k, k, 1, k // 2, up=up)
    M = B * H * W
    out = torch.empty(M, Cout, device=dev)
    op = ops.empty_pair((M, Cout), lib.PREC_BF16X3, dev)
    mask = torch.zeros(M, Cout, dtype=torch.uint8, device=dev)
    resd = res.permute(0, 2, 3, 1).reshape(M, Cout).contiguous().to(dev)
    cv.conv_gemm(_cl_pair(xs, dev), ge, cv.pack_weight(w.to(dev), 0, lib.PREC_BF16X3), Cout, bias=b.to(dev), act=lib.ACT_RELU,
                 residual=resd, out_f32=out, out=op, out_mask=mask)
    torch.cuda.synchronize()
    got = out.cpu().reshape(B, H, W, Cout).permute(0, 3, 1, 2)
    assert rel_l2(got.numpy(), ref.numpy()) < 5e-5
    assert rel_l2((op[0].float() + op[1].float()).cpu().numpy(), out.cpu().numpy()) < 2e-5
    m = mask.cpu().reshape(B, H, W, Cout).permute(0, 3, 1, 2).bool()
    assert (m != ref_mask).float().mean() < 1e-4


@pytest.mark.parametrize("cfg", [
    dict(B=2, H=12, W=10, Cin=128, Cout=128, k=3, up=0),
    dict(B=1, H=16, W=16, Cin=128, Cout=256, k=3, up=2),
    dict(B=2, H=14, W=14, Cin=256, Cout=128, k=1, up=0),
    dict(B=2, H=10, W=12, Cin=128, Cout=4, k=3, up=0),
])
def test_conv_backward_data_and_weight(dev, cfg):
    from mvp import conv as cv, lib, ops

    g = torch.Generator().manual_seed(cfg["Cin"] * 3 + cfg["Cout"])
    B, H, W, Cin, Cout, k, up = (cfg[n] for n in ("B", "H", "W", "Cin", "Cout", "k", "up"))
    xs = torch.randn(B, Cin, H >> up, W >> up, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.05
    gy = torch.randn(B, Cout, H, W, generator=g)
    xs_r = xs.clone().double().requires_grad_(True)
    w_r = w.clone().double().requires_grad_(True)
    x = F.interpolate(xs_r, scale_factor=2 ** up) if up else xs_r
    x.retain_grad()
    F.conv2d(x, w_r, None, padding=k // 2).backward(gy.double())
    M = B * H * W
    ge = cv.geom(B, H, W, Cin, k, k, 1, k // 2, up=up)
    LG = (Cout + 127) // 128 * 128
    gcl = gy.permute(0, 2, 3, 1).reshape(M, Cout).contiguous().to(dev)
    gP = cv.mask_split(gcl, None, M, Cout, ldo=LG)
    # weight gradient
    dw = torch.full((Cout, Cin, k, k), float("nan"), device=dev)
    cv.conv_dw(gP, LG, _cl_pair(xs, dev), Cin, ge, Cout, dw)
    # data gradient (w.r.t. the virtual, upsampled input)
    gd = cv.geom(B, H, W, LG, k, k, 1, k // 2)
    dx = torch.empty(M, Cin, device=dev)
    cv.conv_gemm(gP, gd, cv.pack_weight(w.to(dev), 1, lib.PREC_BF16X3, pad_cout_to=LG), Cin, out_f32=dx)
    torch.cuda.synchronize()
    assert rel_l2(dw.cpu().numpy(), w_r.grad.numpy()) < 5e-5
    assert rel_l2(dx.cpu().reshape(B, H, W, Cin).permute(0, 3, 1, 2).numpy(), x.grad.numpy()) < 5e-5
    if up:
        gs, _ = cv.upsample_nearest(dx, B, H >> up, W >> up, Cin, 2 ** up, want_pair=False, backward=True)
        torch.cuda.synchronize()
        assert rel_l2(gs.cpu().reshape(B, H >> up, W >> up, Cin).permute(0, 3, 1, 2).numpy(), xs_r.grad.numpy()) < 5e-5


class _MaskedRelu(torch.autograd.Function):
    """relu whose backward gate is a GIVEN mask (the HIP path's own byte mask)."""

    @staticmethod
    def forward(c, x, m):
        c.save_for_backward(m)
        return x.relu()

    @staticmethod
    def backward(c, g):
        return g * c.saved_tensors[0], None


def _dpt_fp64_with_masks(sd, feats, ctx, B, h, w):
    """fp64 restatement of probes.py:377-399 (transformer variant) whose ReLU gates are taken from
    the HIP forward.  A 1e-5-accurate forward flips ~1e-5 of the masks of a 16-conv ReLU stack, which
    alone moves parameter gradients by sqrt(1e-5) ~ 3e-3 rel-L2 (measured stage by stage against an fp64 restatement);
    with the gates pinned, the hand-written backward chain must agree to rounding."""
    from mvp import dpt as mdpt

    H1, W1, H2, W2 = 2 * h, 2 * w, 8 * h, 8 * w
    cl = lambda m, H, W: m.cpu().reshape(B, H, W, -1).permute(0, 3, 1, 2).double()
    P = {n: t.clone().double().requires_grad_(True) for n, t in sd.items()}
    f = [F.interpolate(F.conv2d(feats[i].double(), P[f"head.conv_{i}.weight"], P[f"head.conv_{i}.bias"]), scale_factor=2) for i in range(4)]

    def rcu(x, pre, n):
        _, _, ma, mb = ctx.saved_rcu[n][:4]
        a = _MaskedRelu.apply(F.conv2d(x, P[pre + "conv.0.weight"], P[pre + "conv.0.bias"], padding=1), cl(ma, H1, W1))
        return _MaskedRelu.apply(F.conv2d(a, P[pre + "conv.2.weight"], P[pre + "conv.2.bias"], padding=1), cl(mb, H1, W1)) + x

    out = None
    for n, (blk, unit) in enumerate(mdpt.RCU_ORDER):
        pre = f"head.ref_{blk}.resConfUnit{unit}."
        if unit == 1:
            out = rcu(f[blk], pre, n) + out
        elif blk == 3:
            out = rcu(f[3], pre, n)
        else:
            out = rcu(out, pre, n)
    out = F.interpolate(out, scale_factor=4)
    h0 = _MaskedRelu.apply(F.conv2d(out, P["head.out_conv.0.weight"], P["head.out_conv.0.bias"], padding=1), cl(ctx.m0, H2, W2))
    return F.conv2d(h0, P["head.out_conv.2.weight"], P["head.out_conv.2.bias"], padding=1), P


@pytest.mark.parametrize("odim", [256, 4, 1])
def test_dpt_backward_chain_with_pinned_gates(dev, odim):
    from evals.models.probes import DPT
    from mvp import dpt as mdpt
    from mvp import functional as MF
    from oracle import probes as oprobes

    C, Hd, B, h, w = 128, 128, 2, 5, 6
    g = torch.Generator().manual_seed(70 + odim)
    feats = [torch.randn(B, C, h, w, generator=g) for _ in range(4)]
    head = DPT([C] * 4, odim, Hd, 3)
    sd = oprobes.make_dpt_weights([C] * 4, odim, hidden=Hd, k=3, seed=5)
    head.load_state_dict({k[len("head."):]: v for k, v in sd.items()}, strict=True)
    head = head.to(dev)
    pack = MF.pack_features([f.to(dev) for f in feats], head.precision)
    lq = mdpt.dpt_vit_logits(pack, head, head.precision)
    gy = torch.randn(B, 8 * h, 8 * w, odim, generator=g)
    gyp = torch.zeros(lq.shape)
    gyp[..., :odim] = gy
    (lq * gyp.to(dev)).sum().backward()
    torch.cuda.synchronize()
    y, P = _dpt_fp64_with_masks(sd, feats, lq.grad_fn, B, h, w)
    (y * gy.permute(0, 3, 1, 2).double()).sum().backward()
    assert rel_l2(lq.detach().cpu()[..., :odim].permute(0, 3, 1, 2).numpy(), y.detach().numpy()) < 5e-5
    for n, p in head.named_parameters():
        assert rel_l2(p.grad.cpu().numpy(), P["head." + n].grad.numpy()) < 1e-4, n


@pytest.mark.parametrize("kind", ["bindepth", "sigdepth", "snorm_ua"])
def test_dpt_probe_fwd_bwd_vs_oracle(dev, kind):
    from evals.models.probes import DepthHead, SurfaceNormalHead
    from oracle import probes as oprobes

    C, Hd, B, h, w = 128, 128, 2, 5, 6
    g = torch.Generator().manual_seed(77)
    feats = [torch.randn(B, C, h, w, generator=g) for _ in range(4)]
    if kind == "snorm_ua":
        probe = SurfaceNormalHead(feat_dim=[C] * 4, head_type="dpt", uncertainty_aware=True, hidden_dim=Hd, kernel_size=3)
        odim = 4
    else:
        probe = DepthHead(feat_dim=[C] * 4, head_type="dpt", prediction_type=kind, hidden_dim=Hd, kernel_size=3)
        odim = 256 if kind == "bindepth" else 1
    sd = oprobes.make_dpt_weights([C] * 4, odim, hidden=Hd, k=3, seed=5)
    probe.load_state_dict(sd, strict=True)
    probe = probe.to(dev)
    y = probe([f.to(dev) for f in feats])
    sd_r = {n: t.clone().requires_grad_(True) for n, t in sd.items()}
    y_ref = oprobes.snorm_head(sd_r, feats, "dpt", 3) if kind == "snorm_ua" else oprobes.depth_head(sd_r, feats, "dpt", 3, kind)
    assert tuple(y.shape) == tuple(y_ref.shape) == (B, y_ref.shape[1], 16 * h, 16 * w)
    gy = torch.randn(y_ref.shape, generator=g)
    (y_ref * gy).sum().backward()
    (y * gy.to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert rel_l2(y.detach().cpu().numpy(), y_ref.detach().numpy()) < 1e-4
    worst = 0.0
    for n, p in probe.named_parameters():
        e = rel_l2(p.grad.cpu().numpy(), sd_r[n].grad.numpy())
        worst = max(worst, e)
        # vs the plain oracle the ReLU gates differ on ~1e-5 of the elements (see
        # test_dpt_backward_chain_with_pinned_gates): loose rel-L2 bound + direction check
        assert e < 5e-2, (n, e)
        a64, b64 = p.grad.double().cpu().flatten(), sd_r[n].grad.double().flatten()
        assert 1 - float(a64 @ b64 / (a64.norm() * b64.norm())) < 2e-3, n
    print(f"\n[dpt {kind}] worst param-grad rel-L2 = {worst:.2e}")


def test_dpt_probe_trained_in_bf16_tracks_the_three_product_probe_within_the_depth_rmse_tolerance(dev):
    """`precision="bf16"` for the TRAINED probe (`bench.py --probe-precision bf16`: ordinary mixed-precision training — one bf16 MFMA product per
    convolution, fp32 accumulation, fp32 master weights and AdamW — on exact frozen features) is a secondary, opt-in mode; the parity tests above
    hold the default three-product probe to the reference.  What BASELINE.json's north star asks of the depth output is 1e-2 on depth RMSE: from one
    initialisation, on the same features and targets, 12 training steps in either arithmetic end with depth RMSEs within 1e-2 (relative) of each
    other, step by step, and with loss trajectories within 1e-2.  (Reference: train_depth.py:99-144, probes.py:215-399.)"""
    from evals.models.probes import DepthHead
    from evals.utils.losses import DepthLoss
    from mvp import functional as MF
    from mvp.optim import FlatAdamW
    from oracle import probes as oprobes

    C, Hd, B, h, w = 128, 128, 4, 5, 6
    g = torch.Generator().manual_seed(123)
    batches = []
    for _ in range(3):
        feats = [torch.randn(B, C, h, w, generator=g).to(dev) for _ in range(4)]
        tgt = (torch.rand(B, 1, 16 * h, 16 * w, generator=g) * 9.9 + 0.05)
        tgt[torch.rand(tgt.shape, generator=g) < 0.1] = 0.0
        batches.append((feats, tgt.to(dev)))
    sd = oprobes.make_dpt_weights([C] * 4, 256, hidden=Hd, k=3, seed=5)
    out = {}
    for prec in ("bf16x3", "bf16"):
        probe = DepthHead(feat_dim=[C] * 4, head_type="dpt", prediction_type="bindepth", hidden_dim=Hd, kernel_size=3, precision=prec)
        probe.load_state_dict(sd, strict=True)
        probe = probe.to(dev)
        opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}])
        loss_fn = DepthLoss()
        losses, rmses = [], []
        for step in range(12):
            feats, tgt = batches[step % 3]
            opt.zero_grad()
            pred = MF.interpolate(probe(feats), size=tgt.shape[-2:], mode="bilinear")
            loss = loss_fn(pred, tgt.clone())
            loss.backward()
            opt.step()
            valid = tgt > 0
            rmses.append(float(((pred.detach() - tgt)[valid] ** 2).mean().sqrt()))
            losses.append(float(loss))
        out[prec] = (np.array(losses), np.array(rmses))
    torch.cuda.synchronize()
    (la, ra), (lb, rb) = out["bf16x3"], out["bf16"]
    print(f"\n[dpt probe bf16 vs bf16x3] max rel diff: loss {np.abs(lb / la - 1).max():.2e}, depth RMSE {np.abs(rb / ra - 1).max():.2e}; RMSE {ra[0]:.4f} -> {ra[-1]:.4f}")
    assert np.isfinite(lb).all() and la[-3:].mean() < la[:3].mean() and lb[-3:].mean() < lb[:3].mean()  # both train (the scale-invariant loss falls)
    assert np.abs(rb / ra - 1).max() < 1e-2 and np.abs(lb / la - 1).max() < 1e-2


@pytest.mark.parametrize("pt", ["bindepth", "sigdepth"])
def test_linear_probe_k3_vs_oracle(dev, pt):
    """probes.py:417-432 with kernel_size=3 (no conv/resample commutation: explicit bilinear x4 + 3x3 conv)."""
    from evals.models.probes import DepthHead
    from oracle import probes as oprobes

    C, B, h, w = 32, 2, 5, 7
    g = torch.Generator().manual_seed(9)
    feats = [torch.randn(B, C, h, w, generator=g) for _ in range(4)]
    odim = 256 if pt == "bindepth" else 1
    probe = DepthHead(feat_dim=[C] * 4, head_type="linear", kernel_size=3, prediction_type=pt)
    assert probe.name == f"{pt}_linear_k3"
    sd = oprobes.make_linear_head_weights([C] * 4, odim, 3, seed=8)
    probe.load_state_dict(sd, strict=True)
    probe = probe.to(dev)
    y = probe([f.to(dev) for f in feats])
    sd_r = {n: t.clone().requires_grad_(True) for n, t in sd.items()}
    y_ref = oprobes.depth_head(sd_r, feats, "linear", 3, pt)
    gy = torch.randn(y_ref.shape, generator=g)
    (y_ref * gy).sum().backward()
    (y * gy.to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert rel_l2(y.detach().cpu().numpy(), y_ref.detach().numpy()) < 2e-5
    for n, p in probe.named_parameters():
        assert rel_l2(p.grad.cpu().numpy(), sd_r[n].grad.numpy()) < (5e-3 if pt == "bindepth" else 1e-4), n


@pytest.mark.parametrize("f,shape", [(4, (2, 5, 7, 128)), (2, (1, 6, 4, 64)), (4, (1, 1, 1, 8))])
def test_upconv_boxsum_folds_the_backward_of_upsample_then_conv_onto_the_coarse_grid(dev, f, shape):
    """mvp_upconv3_grad_boxsum: with G = box sums of the fine-grid gradient per coarse pixel and tap, the gradients of
    y = conv3x3(interpolate(x, scale_factor=f, mode='nearest'), W, padding=1) (probes.py:396-397) are dW = Gᵀ·x and dx = G·Wᵀ over the
    coarse pixels.  Checked against torch autograd in fp64 on the same graph (image borders included)."""
    import torch.nn.functional as F
    from mvp import conv as cv

    B, H, W, C = shape
    g = torch.Generator().manual_seed(3)
    gy = torch.randn(B, H * f, W * f, C, generator=g)
    G = cv.upconv3_grad_boxsum(gy.to(dev), B, H, W, C, f, precision=3)
    Gf = (G[0].float() + G[1].float()).double().cpu().view(B * H * W, 9, C)
    Cin = 8
    x = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64).requires_grad_(True)
    w = torch.randn(C, Cin, 3, 3, generator=g, dtype=torch.float64).requires_grad_(True)
    y = F.conv2d(F.interpolate(x, scale_factor=f, mode="nearest"), w, padding=1)
    y.backward(gy.permute(0, 3, 1, 2).double())
    xr = x.detach().permute(0, 2, 3, 1).reshape(B * H * W, Cin)
    dW = torch.einsum("mtc,mi->cit", Gf, xr).reshape(C, Cin, 3, 3)
    dX = torch.einsum("mtc,cit->mi", Gf, w.detach().reshape(C, Cin, 9)).view(B, H, W, Cin).permute(0, 3, 1, 2)
    assert float((dW - w.grad).norm() / w.grad.norm()) < 2e-5   # (the pair carries ~16 mantissa bits)
    assert float((dX - x.grad).norm() / x.grad.norm()) < 2e-5


@pytest.mark.parametrize("f,shape,cout", [(4, (2, 5, 7, 128), 128), (2, (1, 6, 4, 128), 256)])
def test_upconv_forward_from_coarse_tap_products(dev, f, shape, cout):
    """cv.upconv3_forward = relu(conv3x3(interpolate(x, scale_factor=f, mode='nearest'), W, padding=1) + b) (probes.py:396-397) computed as
    per-tap products on the coarse grid + a gather-sum per fine pixel: against torch in fp64 (borders included), pair / fp32 / gate
    mask outputs consistent with each other."""
    import torch.nn.functional as F
    from mvp import conv as cv, lib, ops

    B, H, W, C = shape
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, H, W, C, generator=g)
    w = torch.randn(cout, C, 3, 3, generator=g) * 0.05
    b = torch.randn(cout, generator=g)
    xP = ops.split_bf16(x.reshape(-1, C).to(dev), 3)
    M = B * H * f * W * f
    oP = ops.empty_pair((M, cout), 3, dev)
    o32 = torch.empty(M, cout, dtype=torch.float32, device=dev)
    om = torch.empty(M, cout, dtype=torch.uint8, device=dev)
    cv.upconv3_forward(xP, w.to(dev), b.to(dev), B, H, W, f, act=lib.ACT_RELU, out=oP, out_mask=om, out_f32=o32, precision=3)
    ref = F.conv2d(F.interpolate(x.permute(0, 3, 1, 2).double(), scale_factor=f, mode="nearest"), w.double(), b.double(), padding=1).relu()
    ref = ref.permute(0, 2, 3, 1).reshape(M, cout)
    got = o32.double().cpu()
    assert float((got - ref).norm() / ref.norm()) < 2e-5
    assert torch.equal(om.cpu().bool(), got > 0)
    pair = (oP[0].float() + oP[1].float()).double().cpu()
    assert float((pair - got).abs().max()) < 2e-4 * float(got.abs().max())
