"""Full-size property tests (BASELINE.json sizes; no oracle at these sizes — the CPU restatement would take minutes):
size-independent identities the domain offers, checked on the HIP path through the C ABI.

  * attention: softmax rows sum to 1  (V = 1 -> O = 1), and O is a convex combination (min V <= O <= max V)
  * GEMM: linearity in A at the fc1 / fc2 shapes
  * resize: adjoint identity <R x, y> = <x, R^T y> for the three modes, planar and channels-last
  * tap BatchNorm: per-channel output moments (0, 1) and running-stat update
  * backbone: a sample's features do not depend on its batch neighbours (add_norm=False), bit for bit
  * train step at B=16, 224^2: bit-reproducible (every reduction in the path has a fixed order), loss finite, AdamW moves weights
  * SPair at 800^2 (50x50 tokens): an image corresponds to itself (argmax = the query cell)
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


@pytest.mark.parametrize("B,N", [(16, 197), (2, 1201), (1, 2501)])
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_attention_rows_sum_to_one(dev, B, N, precision):
    from mvp import lib, ops
    from mvp.vit import parse_precision

    pr = parse_precision(precision)
    H = 12
    g = torch.Generator(device="cpu").manual_seed(N)
    qkv = torch.randn(B * N, 3 * H * 64, generator=g) * 2.0
    qkv[:, 2 * H * 64:] = 1.0  # V = 1
    out = ops.empty_pair((B * N, H * 64), lib.PREC_BF16X3, dev)
    ops.attention(ops.split_bf16(qkv.to(dev), lib.PREC_BF16X3), out, B, N, H, 0.125, pr)
    o = out[0].float() + out[1].float()
    assert torch.isfinite(o).all()
    assert (o - 1.0).abs().max().item() < (2e-5 if pr == lib.PREC_BF16X3 else 5e-3)
    # convex combination: random V in [-1, 3] -> outputs stay inside the hull
    v = torch.rand(B * N, H * 64, generator=g) * 4 - 1
    qkv[:, 2 * H * 64:] = v
    ops.attention(ops.split_bf16(qkv.to(dev), lib.PREC_BF16X3), out, B, N, H, 0.125, pr)
    o = out[0].float() + out[1].float()
    assert o.min().item() >= -1 - 2e-2 and o.max().item() <= 3 + 2e-2


@pytest.mark.parametrize("shape", [(3152, 3072, 768), (3152, 768, 3072), (19216, 2304, 768)])
def test_gemm_linearity_full_size(dev, shape):
    from mvp import lib, ops

    M, N, K = shape
    g = torch.Generator(device="cpu").manual_seed(M + N)
    a1, a2 = torch.randn(M, K, generator=g).to(dev), torch.randn(M, K, generator=g).to(dev)
    w = ops.split_bf16((torch.randn(N, K, generator=g) * 0.05).to(dev))
    outs = []
    for a in (a1, a2, a1 + a2):
        o = torch.empty(M, N, device=dev)
        ops.gemm(ops.split_bf16(a), w, M, N, K, out_f32=o)
        outs.append(o)
    err = (outs[2] - outs[0] - outs[1]).norm() / outs[2].norm()
    assert err.item() < 2e-5, err.item()
    # and the same launch twice is bit-identical
    o2 = torch.empty(M, N, device=dev)
    ops.gemm(ops.split_bf16(a1 + a2), w, M, N, K, out_f32=o2)
    assert torch.equal(o2, outs[2])


@pytest.mark.parametrize("mode", ["nearest", "bilinear", "bicubic"])
@pytest.mark.parametrize("geom", [(16, 56, 56, 224, 224), (4, 120, 160, 480, 640), (3, 224, 224, 56, 56)])
def test_resize_adjoint_identity(dev, mode, geom):
    from mvp import lib, ops

    planes, Hi, Wi, Ho, Wo = geom
    m = {"nearest": lib.RESIZE_NEAREST, "bilinear": lib.RESIZE_BILINEAR, "bicubic": lib.RESIZE_BICUBIC}[mode]
    g = torch.Generator(device="cpu").manual_seed(Hi + Wo)
    x = torch.randn(planes, Hi, Wi, generator=g).to(dev)
    y = torch.randn(planes, Ho, Wo, generator=g).to(dev)
    rx, rty = torch.empty(planes, Ho, Wo, device=dev), torch.empty(planes, Hi, Wi, device=dev)
    ops.resize(x, rx, planes, Hi, Wi, Ho, Wo, m)
    ops.resize(y, rty, planes, Hi, Wi, Ho, Wo, m, backward=True)
    lhs, rhs = (rx.double() * y.double()).sum().item(), (x.double() * rty.double()).sum().item()
    assert abs(lhs - rhs) < 1e-5 * (abs(lhs) + (rx.double().norm() * y.double().norm()).item() * 1e-2 + 1.0), (lhs, rhs)
    # channels-last with C = 8 must agree with the planar kernels
    C = 8
    Bn = max(planes // C, 1)
    xc = torch.randn(Bn, Hi, Wi, C, generator=g).to(dev)
    yc = torch.randn(Bn, Ho, Wo, C, generator=g).to(dev)
    rxc, rtyc = torch.empty(Bn, Ho, Wo, C, device=dev), torch.empty(Bn, Hi, Wi, C, device=dev)
    ops.resize(xc, rxc, Bn, Hi, Wi, Ho, Wo, m, channels_last=True, Cdim=C)
    ops.resize(yc, rtyc, Bn, Hi, Wi, Ho, Wo, m, channels_last=True, Cdim=C, backward=True)
    lhs, rhs = (rxc.double() * yc.double()).sum().item(), (xc.double() * rtyc.double()).sum().item()
    assert abs(lhs - rhs) < 1e-5 * (abs(lhs) + (rxc.double().norm() * yc.double().norm()).item() * 1e-2 + 1.0), (lhs, rhs)
    xp = xc.permute(0, 3, 1, 2).contiguous()
    rp = torch.empty(Bn * C, Ho, Wo, device=dev)
    ops.resize(xp, rp, Bn * C, Hi, Wi, Ho, Wo, m)
    assert torch.allclose(rp.view(Bn, C, Ho, Wo).permute(0, 2, 3, 1), rxc, rtol=0, atol=2e-5)  # different (valid) association orders


def test_tap_bn_moments_full_size(dev):
    from mvp import ops

    B, N, C, hw = 16, 197, 768, 196
    g = torch.Generator(device="cpu").manual_seed(5)
    x = (torch.randn(B * N, C, generator=g) * torch.linspace(0.1, 30.0, C) + torch.linspace(-50, 50, C)).to(dev)
    ws = torch.empty(ops.bn_tokens_workspace_bytes(B * N, C) // 4, dtype=torch.float32, device=dev)
    stats = torch.empty(2 * C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    nchw = torch.empty(B, C, 14, 14, device=dev)
    ops.bn_tokens_to_nchw(x, B, N, C, hw, workspace=ws, stats=stats, running_mean=rm, running_var=rv, nchw=nchw)
    xd = x.double()
    mean, var = xd.mean(0), xd.var(0, unbiased=False)
    assert torch.allclose(stats[:C].double(), mean, rtol=1e-6, atol=1e-6)
    assert torch.allclose(stats[C:].double(), var, rtol=1e-5)
    assert torch.allclose(rm.double(), 0.1 * mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(rv.double(), 0.9 + 0.1 * xd.var(0, unbiased=True), rtol=1e-5)
    # the spatial tokens (CLS dropped) normalised with the all-token statistics
    ref = ((xd.view(B, N, C)[:, 1:] - mean) / (var + 1e-5).sqrt()).permute(0, 2, 1).reshape(B, C, 14, 14)
    assert (nchw.double() - ref).abs().max().item() < 2e-4


def test_backbone_sample_independent_of_batch(dev):
    from evals.models.dino import DINO
    from oracle import vit as ovit

    vsd = ovit.make_vit_weights(seed=1)
    model = DINO(return_multilayer=True, add_norm=False, weights=vsd).to(dev).eval()
    g = torch.Generator(device="cpu").manual_seed(11)
    imgs = torch.randn(16, 3, 224, 224, generator=g).to(dev)
    full = [f.clone() for f in model(imgs)]
    one = [f.clone() for f in model(imgs[5:6].contiguous())]
    for a, b in zip(full, one):
        assert a.shape == (16, 768, 14, 14) and torch.isfinite(a).all()
        assert torch.equal(a[5:6], b), "a sample's features must not depend on the other images of the batch"


def test_full_size_step_is_reproducible(dev):
    from evals.models.dino import DINO
    from evals.models.probes import DepthHead
    from evals.utils.losses import DepthLoss
    from mvp.optim import FlatAdamW
    from mvp.train import train_depth_step
    from oracle import train as otrain
    from oracle import vit as ovit

    vsd = ovit.make_vit_weights(seed=2)
    images, tgt = otrain.synthetic_depth_batch(16, 224, 224, rank=0, step=0)
    images, tgt = images.to(dev), tgt.to(dev)
    runs = []
    for _ in range(2):
        torch.manual_seed(0)
        model = DINO(return_multilayer=True, add_norm=True, weights=vsd).to(dev)
        probe = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type="bindepth").to(dev)
        opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}])
        w0 = probe.head.conv.weight.detach().clone()
        losses = [train_depth_step(model, probe, opt, None, DepthLoss(), images, tgt.clone()).item() for _ in range(3)]
        runs.append((losses, probe.head.conv.weight.detach().clone(), model.batchnorms[3].running_var.clone()))
        assert all(np.isfinite(losses)) and not torch.equal(w0, runs[-1][1])
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])


def test_spair_self_correspondence_800(dev):
    """BASELINE config #5 shape: 800x800 -> 50x50 tokens of 768 channels; keypoints at cell centres of a random map
    must map to their own cell when source = target."""
    from mvp import spair

    g = torch.Generator(device="cpu").manual_seed(3)
    f = torch.randn(768, 50, 50, generator=g).to(dev)
    K = 20
    cols, rows = torch.randint(0, 50, (K,), generator=g), torch.randint(0, 50, (K,), generator=g)
    # align_corners=True grid_sample convention of the reference (evaluate_spair_correspondence.py:59-66): x01 = col / (w - 1)
    kp = torch.stack((cols.float() / 49.0, rows.float() / 49.0), dim=1)
    xy, val = spair.correspondence(f, f, kp)
    assert torch.equal(xy.cpu(), torch.stack((cols, rows), dim=1))
    assert (val.cpu() - 1.0).abs().max().item() < 1e-4


def test_trajectory_independent_of_host_syncs(dev):
    """The host may run many steps ahead of the stream (bench.py never syncs inside the timed loop): nothing on the path
    may read host-staged state late.  6 steps with a sync after every step == 6 steps enqueued back to back, bit for bit
    (this caught an AdamW schedule buffer that was staged through pinned memory)."""
    from evals.models.dino import DINO
    from evals.models.probes import DepthHead
    from evals.utils.losses import DepthLoss
    from evals.utils.optim import cosine_decay_linear_warmup
    from mvp.optim import FlatAdamW
    from mvp.train import train_depth_step
    from oracle import train as otrain
    from oracle import vit as ovit

    vsd = ovit.make_vit_weights(seed=2)
    batches = []
    for s in range(3):
        im, tg = otrain.synthetic_depth_batch(8, 224, 224, rank=0, step=s)
        batches.append((im.to(dev), tg.to(dev)))
    finals = []
    for sync in (True, False):
        torch.manual_seed(0)
        model = DINO(return_multilayer=True, add_norm=True, weights=vsd).to(dev)
        probe = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type="bindepth").to(dev)
        opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-3}])
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda e: cosine_decay_linear_warmup(e, 20, 4))
        losses = []
        for i in range(6):
            im, tg = batches[i % 3]
            losses.append(train_depth_step(model, probe, opt, sched, DepthLoss(), im, tg.clone()))
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        finals.append((torch.stack(losses).cpu(), probe.head.conv.weight.detach().clone(), probe.head.conv.bias.detach().clone()))
    assert torch.equal(finals[0][0], finals[1][0])
    assert torch.equal(finals[0][1], finals[1][1]) and torch.equal(finals[0][2], finals[1][2])


def test_probe_training_reduces_loss_on_a_learnable_task(dev):
    """Behavioural check of the whole loop (beyond parity): a linear bindepth probe on a frozen tiny ViT, trained for 150 steps on
    images whose depth target is a smooth function of the image itself, must bring DepthLoss down substantially."""
    import torch.nn.functional as F
    from evals.models.dino import DINO
    from evals.models.probes import DepthHead
    from evals.utils.losses import DepthLoss
    from evals.utils.optim import cosine_decay_linear_warmup
    from mvp.optim import FlatAdamW
    from mvp.train import train_depth_step
    from oracle import vit as ovit

    vsd = ovit.make_vit_weights(embed_dim=128, depth=4, seed=5)
    model = DINO(return_multilayer=True, add_norm=True, weights=vsd).to(dev)
    torch.manual_seed(0)
    probe = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type="bindepth").to(dev)
    opt = FlatAdamW([{"params": probe.parameters(), "lr": 2e-3}])
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda e: cosine_decay_linear_warmup(e, 150, 10))
    g = torch.Generator().manual_seed(1)
    batches = []
    for _ in range(8):
        img = torch.randn(4, 3, 64, 64, generator=g)
        depth = 5.0 + 3.0 * torch.tanh(F.avg_pool2d(img.mean(1, keepdim=True), 16).repeat_interleave(16, 2).repeat_interleave(16, 3))  # in (2, 8)
        batches.append((img.to(dev), depth.to(dev)))
    loss_fn = DepthLoss()
    losses = []
    for s in range(150):
        img, tgt = batches[s % len(batches)]
        losses.append(train_depth_step(model, probe, opt, sched, loss_fn, img, tgt.clone()))
    losses = torch.stack(losses).cpu()
    first, last = losses[:8].mean().item(), losses[-8:].mean().item()
    assert torch.isfinite(losses).all() and last < 0.6 * first, (first, last)
