"""GPU parity of the round-2 additions: scale-invariant training branch, antialiased Resize, MultiscaleHead, DINO return_cls,
the feature-pack cache, all against reference-generated goldens or the golden-pinned oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_scale_invariant_branch_vs_reference_golden(dev):
    """train_depth.py:114-118 with the reference's own match_scale_and_shift + clamp + DepthLoss (golden si_train.npz)."""
    from evals.utils.losses import DepthLoss
    from evals.utils.metrics import match_scale_and_shift

    g = load_golden("si_train.npz")
    pred = torch.from_numpy(g["pred"]).to(dev).requires_grad_(True)
    tgt = torch.from_numpy(g["target"]).to(dev)
    m = match_scale_and_shift(pred, tgt)
    np.testing.assert_allclose(m.detach().cpu().numpy(), g["matched"], rtol=5e-5, atol=5e-6)
    c = match_scale_and_shift(pred, tgt, clamp=(0.001, 1.0))
    np.testing.assert_allclose(c.detach().cpu().numpy(), g["clamped"], rtol=5e-5, atol=5e-6)
    loss = DepthLoss()(c, tgt.clone())
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    assert rel_l2(pred.grad.cpu().numpy(), g["grad"]) < 1e-3
    # pixels that the clamp saturates carry exactly zero gradient (torch.clamp semantics)
    sat = (g["matched"] < 0.001) | (g["matched"] > 1.0)
    assert sat.any() and np.all(pred.grad.cpu().numpy()[sat] == 0)


def test_scale_invariant_train_step_vs_oracle(dev):
    from evals.models.dino import DINO
    from evals.models.probes import DepthHead
    from evals.utils.losses import DepthLoss
    from mvp.optim import FlatAdamW
    from mvp.train import train_depth_step
    from oracle import probes as oprobes, train as otrain, vit as ovit

    D = 128
    vsd = ovit.make_vit_weights(embed_dim=D, depth=4, seed=71)
    psd = oprobes.make_linear_head_weights([D] * 4, 1, 1, seed=72)
    images, tgt = otrain.synthetic_depth_batch(3, 64, 80, rank=0, step=0)
    tgt = tgt / 10.0  # relative depth in (0, 1]
    ref = otrain.DepthProbeTrainer(vsd, psd, layers=(0, 1, 2, 3), heads=2, prediction_type="sigdepth", max_depth=1, max_step=20, warmup_step=2,
                                   scale_invariant=True)
    loss_ref = ref.step(images, tgt.clone())
    model = DINO(return_multilayer=True, add_norm=True, weights=vsd).to(dev)
    probe = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type="sigdepth", max_depth=1)
    probe.load_state_dict(psd, strict=True)
    probe = probe.to(dev)
    opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4 * 0.01}])  # lambda(0) = 0.01 of the oracle's schedule
    loss = train_depth_step(model, probe, opt, None, DepthLoss(), images.to(dev), tgt.to(dev), scale_invariant=True)
    assert abs(loss.item() - loss_ref) < 2e-4 * abs(loss_ref)
    assert rel_l2(probe.head.conv.weight.detach().cpu().numpy(), ref.probe_sd["head.conv.weight"].detach().numpy()) < 1e-4


@pytest.mark.parametrize("shape,size", [((2, 3, 480, 640), 480), ((1, 3, 530, 300), 224), ((2, 3, 96, 150), 64), ((1, 2, 64, 64), 96)])
def test_antialiased_resize_vs_torch(dev, shape, size):
    """transforms.Resize((S,S)) on a tensor = F.interpolate(bilinear, antialias=True) (dino_res50.py:80,85): one axis may shrink
    while the other grows or stays (NYU 480x640 -> 480x480)."""
    from mvp import functional as MF

    x = torch.randn(*shape, generator=torch.Generator().manual_seed(1))
    ref = F.interpolate(x, size=(size, size), mode="bilinear", align_corners=False, antialias=True)
    y = MF.resize_antialias(x.to(dev), (size, size))
    assert y.shape == ref.shape
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=2e-6)


def test_resnet_wrapper_downsamples_with_antialias(dev):
    """DINO_RESNET on an input wider than fixed_size (the NYU no-crop case) vs the oracle's antialiased Resize."""
    from evals.models.dino_res50 import DINO_RESNET
    from oracle import resnet as ores

    sd = ores.make_resnet50_weights(seed=8)
    images = torch.randn(2, 3, 96, 128, generator=torch.Generator().manual_seed(2))
    m = DINO_RESNET(return_layers=[1, 2, 3, 4], return_multilayer=False, add_norm=False, fixed_size=96, weights=sd).to(dev)
    out = m(images.to(dev))
    ref = ores.resnet_dense_features(sd, images, [4], fixed_size=96, add_norm=False)
    assert tuple(out.shape) == tuple(ref.shape) == (2, 2048, 3, 3)
    assert rel_l2(out.cpu().numpy(), ref.numpy()) < 1e-3


@pytest.mark.parametrize("kind", ["depth_bins_vit", "snorm_ua_vit", "depth_sig_pyramid"])
def test_multiscale_head_vs_oracle(dev, kind):
    """MultiscaleHead (probes.py:435-458) forward + all parameter gradients vs the oracle (pinned to the reference module by
    tests/golden/probes_multiscale.npz).  The HIP path commutes the 1x1 convs with the bilinear resamples (see mvp/multiscale.py)."""
    from evals.models.probes import DepthHead, SurfaceNormalHead
    from oracle import probes as oprobes

    g = torch.Generator().manual_seed(5)
    B, Hd = 2, 128
    if kind == "depth_sig_pyramid":
        dims = [128, 256, 128, 128]
        feats = [torch.randn(B, dims[i], 3 * 2 ** (3 - i), 4 * 2 ** (3 - i), generator=g) for i in range(4)]
    else:
        dims = [128] * 4
        feats = [torch.randn(B, 128, 5, 6, generator=g) for _ in range(4)]
    if kind == "snorm_ua_vit":
        probe, odim = SurfaceNormalHead(feat_dim=dims, head_type="multiscale", uncertainty_aware=True, hidden_dim=Hd, kernel_size=1), 4
    elif kind == "depth_bins_vit":
        probe, odim = DepthHead(feat_dim=dims, head_type="multiscale", prediction_type="bindepth", hidden_dim=Hd, kernel_size=1), 256
    else:
        probe, odim = DepthHead(feat_dim=dims, head_type="multiscale", prediction_type="sigdepth", hidden_dim=Hd, kernel_size=1), 1
    sd = oprobes.make_multiscale_weights(dims, odim, hidden=Hd, k=1, seed=9)
    probe.load_state_dict(sd, strict=True)
    probe = probe.to(dev)
    y = probe([f.to(dev) for f in feats])
    sd_r = {n: t.clone().requires_grad_(True) for n, t in sd.items()}
    if kind == "snorm_ua_vit":
        y_ref = oprobes.snorm_head(sd_r, feats, "multiscale", 1)
    else:
        y_ref = oprobes.depth_head(sd_r, feats, "multiscale", 1, "bindepth" if odim == 256 else "sigdepth")
    assert tuple(y.shape) == tuple(y_ref.shape)
    gy = torch.randn(y_ref.shape, generator=g)
    (y_ref * gy).sum().backward()
    (y * gy.to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert rel_l2(y.detach().cpu().numpy(), y_ref.detach().numpy()) < 1e-4, kind
    for n, p in probe.named_parameters():
        a64, b64 = p.grad.double().cpu().flatten(), sd_r[n].grad.double().flatten()
        assert float((a64 - b64).norm() / b64.norm()) < 3e-2, (kind, n)            # ReLU-gate flips near 0
        assert 1 - float(a64 @ b64 / (a64.norm() * b64.norm())) < 1e-3, (kind, n)


@pytest.mark.parametrize("kind", ["depth_bins", "snorm_ua"])
def test_multiscale_head_k3_vs_oracle(dev, kind):
    """MultiscaleHead(kernel_size=3) (probes.py:435-458 with make_conv's un-padded 3x3 convs, :400-412): forward + all parameter
    gradients vs the oracle (pinned to the reference module for k = 3 by tests/golden/probes_multiscale.npz ``depth_ms_k3``).
    7x8 maps -> 5x6 -> x2 -> three convs -> 4x6 -> x4 -> two convs -> 12x20."""
    from evals.models.probes import DepthHead, SurfaceNormalHead
    from oracle import probes as oprobes

    g = torch.Generator().manual_seed(6)
    B, Hd, dims = 2, 128, [128] * 4
    feats = [torch.randn(B, 128, 7, 8, generator=g) for _ in range(4)]
    if kind == "snorm_ua":
        probe, odim = SurfaceNormalHead(feat_dim=dims, head_type="multiscale", uncertainty_aware=True, hidden_dim=Hd, kernel_size=3), 4
    else:
        probe, odim = DepthHead(feat_dim=dims, head_type="multiscale", prediction_type="bindepth", hidden_dim=Hd, kernel_size=3), 256
    sd = oprobes.make_multiscale_weights(dims, odim, hidden=Hd, k=3, seed=10)
    probe.load_state_dict(sd, strict=True)
    probe = probe.to(dev)
    y = probe([f.to(dev) for f in feats])
    sd_r = {n: t.clone().requires_grad_(True) for n, t in sd.items()}
    y_ref = oprobes.snorm_head(sd_r, feats, "multiscale", 3) if kind == "snorm_ua" else oprobes.depth_head(sd_r, feats, "multiscale", 3, "bindepth")
    assert tuple(y.shape) == tuple(y_ref.shape) and tuple(y.shape[-2:]) == (12, 20)
    gy = torch.randn(y_ref.shape, generator=g)
    (y_ref * gy).sum().backward()
    (y * gy.to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert rel_l2(y.detach().cpu().numpy(), y_ref.detach().numpy()) < 1e-4, kind
    for n, p in probe.named_parameters():
        a64, b64 = p.grad.double().cpu().flatten(), sd_r[n].grad.double().flatten()
        assert float((a64 - b64).norm() / b64.norm()) < 3e-2, (kind, n)            # ReLU-gate flips near 0
        assert 1 - float(a64 @ b64 / (a64.norm() * b64.norm())) < 1e-3, (kind, n)


def test_dino_return_cls(dev):
    """dino.py:206-207: single tap + return_cls -> embeds[0][:, 0] (tap-BN normalised when add_norm)."""
    from evals.models.dino import DINO
    from oracle import vit as ovit

    sd = ovit.make_vit_weights(embed_dim=128, depth=4, seed=81)
    images = torch.randn(3, 3, 64, 96, generator=torch.Generator().manual_seed(4))
    for add_norm in (False, True):
        m = DINO(layer=2, return_cls=True, add_norm=add_norm, weights=sd).to(dev)
        cls = m(images.to(dev))
        tok = ovit.vit_dense_features(sd, images, [2], heads=2, add_norm=add_norm, return_tokens=True)[0]
        assert tuple(cls.shape) == (3, 128)
        assert rel_l2(cls.cpu().numpy(), tok[:, 0].numpy()) < 1e-3, add_norm


def test_feature_pack_cache_is_not_fooled_by_recycled_addresses(dev):
    """after the backbone's maps are freed, fresh same-shape tensors (possibly at the same addresses, version 0) must be
    repacked by the probe, not served from the cached packing of the previous model() call."""
    from evals.models.dino import DINO
    from evals.models.probes import DepthHead
    from mvp import vit as mvit
    from oracle import vit as ovit

    sd = ovit.make_vit_weights(embed_dim=128, depth=4, seed=82)
    model = DINO(return_multilayer=True, add_norm=True, weights=sd).to(dev)
    torch.manual_seed(0)
    probe = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type="bindepth").to(dev)
    g = torch.Generator().manual_seed(6)
    img_a, img_b = torch.randn(2, 3, 64, 64, generator=g).to(dev), torch.randn(2, 3, 64, 64, generator=g).to(dev)
    with torch.no_grad():
        fb = [f.clone() for f in model(img_b)]                  # features of image B, private copies
        ref_b = probe(fb).clone()                               # packed from the tensors themselves
        fa = model(img_a)                                       # registry now holds image A's packing
        shapes = [f.shape for f in fa]
        ptrs = [f.data_ptr() for f in fa]
        del fa
        fresh = [torch.empty(s, dtype=torch.float32, device=dev) for s in shapes]  # may land on the freed addresses
        for t, src in zip(fresh, fb):
            t.copy_(src)
        assert mvit.lookup_pack(fresh) is None
        out = probe(fresh)
    assert torch.equal(out, ref_b)
    assert [t.data_ptr() for t in fresh] != ptrs  # the registry's strong references keep the old maps' storage alive
    # the intended fast path still works: the very tensors the backbone returned hit the cache
    with torch.no_grad():
        fa = model(img_a)
        assert mvit.lookup_pack([f.detach() for f in fa]) is fa.packed
