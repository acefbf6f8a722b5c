"""BASELINE.json configs #2, #3, #4 as composed WORKLOADS (backbone -> probe -> upsample -> loss -> backward -> FlatAdamW through the
product's train_*_step), checked end to end against the golden-pinned CPU oracle at CPU-checkable sizes, plus full-size property
tests of the same compositions (finite, bit-reproducible, index ranges) and a ResNet trunk case large enough to take the 128x128
and 64x128 conv tiles.  Config #1 (ResNet-50 single tap) and #5 (SPair) are in test_gpu_resnet.py / test_gpu_wrappers.py."""
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


def _grad_check(probe, ref_sd, rel=5e-2, cos=2e-3):
    """Parameter gradients: loose rel-L2 + tight cosine (ReLU / |.| gates within rounding of 0 take the other branch)."""
    for n, p in probe.named_parameters():
        a, b = p.grad.double().cpu().flatten(), ref_sd[n].grad.double().flatten()
        assert float((a - b).norm() / b.norm().clamp_min(1e-30)) < rel, n
        assert 1 - float(a @ b / (a.norm() * b.norm()).clamp_min(1e-30)) < cos, n


def _adam_first_step(ref_sd, lr):
    from oracle import optim as ooptim

    names = list(ref_sd)
    m_, v_ = [torch.zeros_like(ref_sd[n]) for n in names], [torch.zeros_like(ref_sd[n]) for n in names]
    with torch.no_grad():
        ooptim.adamw_step([ref_sd[n] for n in names], [ref_sd[n].grad for n in names], m_, v_, 1, lr)


# ---------------------------------------------------------------------------------------------------------------- config #2
def test_config2_dino_vitb16_linear_bindepth_480x640_whole_step(dev):
    """#2: DINO ViT-B/16 return_multilayer (4 taps, train-mode tap BN) + DepthHead(linear, k=1, bindepth) -> bilinear to 480x640 ->
    DepthLoss -> backward -> AdamW, B=1, FULL resolution (N=1201 tokens): loss, prediction RMSE <= 1e-2, gradients, updated weights."""
    from evals.models.dino import DINO
    from evals.models.probes import DepthHead
    from evals.utils.losses import DepthLoss
    from mvp import functional as MF
    from mvp.optim import FlatAdamW
    from mvp.train import train_depth_step
    from oracle import probes as oprobes, train as otrain, vit as ovit

    vsd = ovit.make_vit_weights(seed=0)
    psd = oprobes.make_linear_head_weights([768] * 4, 256, 1, seed=2)
    images, tgt = otrain.synthetic_depth_batch(1, 480, 640, rank=0, step=0)
    ref = otrain.DepthProbeTrainer(vsd, psd, max_step=100, warmup_step=10)
    feats = ref.features(images)
    loss_ref, pred_ref = ref.forward_loss(feats, tgt.clone())
    loss_ref.backward()
    _adam_first_step(ref.probe_sd, ref.lr_at(0))

    model = DINO(return_multilayer=True, add_norm=True, weights=vsd).to(dev)
    probe = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type="bindepth")
    probe.load_state_dict(psd, strict=True)
    probe = probe.to(dev)
    with torch.no_grad():
        pred = MF.interpolate(probe(model(images.to(dev))), size=(480, 640), mode="bilinear")
    p, r = pred.cpu().double(), pred_ref.detach().double()
    assert float(((p - r) ** 2).mean().sqrt() / (r ** 2).mean().sqrt()) < 1e-2            # north_star: 1e-2 on depth RMSE
    assert rel_l2(p.numpy(), r.numpy()) < 1e-3
    opt = FlatAdamW([{"params": probe.parameters(), "lr": ref.lr_at(0)}])
    loss = train_depth_step(model, probe, opt, None, DepthLoss(), images.to(dev), tgt.to(dev))
    assert abs(loss.item() - loss_ref.item()) < 2e-4 * abs(loss_ref.item())
    _grad_check(probe, ref.probe_sd, rel=5e-3, cos=1e-5)   # measured 1.7e-3 / 1.4e-6
    w, w_ref = probe.head.conv.weight.detach().cpu().numpy(), ref.probe_sd["head.conv.weight"].detach().numpy()
    assert rel_l2(w, w_ref) < 2e-3  # first Adam step = -lr*sign(g): near-zero gradients may flip sign (see test_snorm_train_step_vs_oracle)


# ---------------------------------------------------------------------------------------------------------------- config #3
def test_config3_mocov3_resnet50_dpt_snorm_step(dev):
    """#3: MoCoV3_RES(return_layers [1,2,3,4], multilayer, add_norm) -> SurfaceNormalHead(dpt, UA) -> bicubic -> angular_loss ->
    FlatAdamW through train_snorm_step (train_snorm.py:93-120).  Full ResNet-50 widths at 128^2 (pyramid 32/16/8/4), hidden 128."""
    from evals.models.mocov3_res50 import MoCoV3_RES
    from evals.models.probes import SurfaceNormalHead
    from mvp.optim import FlatAdamW
    from mvp.train import train_snorm_step
    from oracle import losses as olosses, probes as oprobes, resnet as ores, train as otrain

    S, B, Hd = 128, 2, 128
    rsd = ores.make_resnet50_weights(seed=13)
    images, depth, normals = otrain.synthetic_snorm_batch(B, S, S, rank=0, step=0)
    mask = depth > 0
    model = MoCoV3_RES(return_layers=[1, 2, 3, 4], return_multilayer=True, add_norm=True, fixed_size=S, weights=rsd).to(dev)
    fd = model.feat_dim
    assert [c for c, _ in fd] == [256, 512, 1024, 2048]
    probe = SurfaceNormalHead(feat_dim=fd, head_type="dpt", uncertainty_aware=True, hidden_dim=Hd, kernel_size=3)
    psd = oprobes.make_dpt_weights(fd, 4, hidden=Hd, k=3, seed=14)
    probe.load_state_dict(psd, strict=True)
    probe = probe.to(dev)
    # oracle
    feats = ores.resnet_dense_features(rsd, images, [1, 2, 3, 4], fixed_size=S)
    p_ref = {k: v.clone().requires_grad_(True) for k, v in psd.items()}
    pred_ref = F.interpolate(oprobes.snorm_head(p_ref, [f.clone() for f in feats], "dpt", 3), size=(S, S), mode="bicubic")
    loss_ref = olosses.angular_loss(pred_ref, normals, mask, uncertainty_aware=True)
    loss_ref.backward()
    # product
    with torch.no_grad():
        fh = model(images.to(dev))
    for o, r in zip(fh, feats):
        assert rel_l2(o.cpu().numpy(), r.numpy()) < 1e-3
    opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}])
    loss = train_snorm_step(model, probe, opt, None, images.to(dev), normals.to(dev), mask.to(dev))
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_ref.item()) < 5e-4 * abs(loss_ref.item())
    _grad_check(probe, p_ref, rel=6e-2, cos=2e-3)   # measured 3.0e-2 / 4.5e-4: ReLU-gate flips through 4 fusion stages


# ---------------------------------------------------------------------------------------------------------------- config #4
def test_config4_mae_vitb16_dpt_bindepth_step(dev):
    """#4: MAE ViT-B/16 (HF key layout, eps 1e-12, taps = block inputs, sincos pos-embed rebuilt by resize_pos_embed as
    train_depth.py:613-617 does) -> DepthHead(dpt, k=3, bindepth) -> bilinear -> DepthLoss -> FlatAdamW; 96x128 input, hidden 128."""
    from evals.models.mae import MAE
    from evals.models.probes import DepthHead
    from evals.utils.losses import DepthLoss
    from mvp import functional as MF
    from mvp.optim import FlatAdamW
    from mvp.train import train_depth_step
    from oracle import losses as olosses, probes as oprobes, train as otrain, vit as ovit
    from test_gpu_wrappers import _to_hf

    H, W, B, Hd = 96, 128, 2, 128
    vsd = ovit.make_vit_weights(seed=21)
    images, tgt = otrain.synthetic_depth_batch(B, H, W, rank=0, step=0)
    model = MAE(return_multilayer=True, add_norm=True, weights=_to_hf(vsd)).to(dev)
    assert model.multilayers == [2, 5, 8, 11]
    model.resize_pos_embed(image_size=(H, W))
    assert (model.feat_h, model.feat_w) == (6, 8)
    probe = DepthHead(feat_dim=model.feat_dim, head_type="dpt", kernel_size=3, prediction_type="bindepth", hidden_dim=Hd)
    psd = oprobes.make_dpt_weights([768] * 4, 256, hidden=Hd, k=3, seed=22)
    probe.load_state_dict(psd, strict=True)
    probe = probe.to(dev)
    # oracle
    sd_ref = dict(vsd)
    sd_ref["pos_embed"] = ovit.sincos_pos_embed_2d(768, (6, 8), True)
    with torch.no_grad():
        feats = ovit.vit_dense_features(sd_ref, images, [2, 5, 8, 11], heads=12, ln_eps=1e-12, pos_mode="fixed", tap_input_of_block=True)
    p_ref = {k: v.clone().requires_grad_(True) for k, v in psd.items()}
    pred_ref = F.interpolate(oprobes.depth_head(p_ref, [f.clone() for f in feats], "dpt", 3, "bindepth"), size=(H, W), mode="bilinear")
    loss_ref = olosses.depth_loss(pred_ref, tgt.clone())
    loss_ref.backward()
    # product
    with torch.no_grad():
        pred = MF.interpolate(probe(model(images.to(dev))), size=(H, W), mode="bilinear")
    p, r = pred.cpu().double(), pred_ref.detach().double()
    assert float(((p - r) ** 2).mean().sqrt() / (r ** 2).mean().sqrt()) < 1e-2
    opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}])
    loss = train_depth_step(model, probe, opt, None, DepthLoss(), images.to(dev), tgt.to(dev))
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_ref.item()) < 5e-4 * abs(loss_ref.item())
    _grad_check(probe, p_ref, rel=3e-2, cos=5e-4)   # measured 9.7e-3 / 4.7e-5


# ------------------------------------------------------------------------------------------------ ResNet trunk, large-M conv tiles
def test_resnet50_trunk_large_tiles_vs_oracle(dev):
    """ResNet-50 at the reference's 480^2, B=2: layer1 runs M = 2*120*120 = 28800 rows (conv3, N=256: 450 tiles of 128x128 -> the
    8-wave 128x128 conv tile, gemm.hip c128 >= 300), layer2 M = 7200 (conv3, N=512: 228 tiles -> the 64x128 tile, c128 >= 160);
    the 96^2 / 128^2 cases never reach either."""
    from evals.models.dino_res50 import DINO_RESNET
    from oracle import resnet as ores

    sd = ores.make_resnet50_weights(seed=5)
    images = torch.randn(2, 3, 480, 480, generator=torch.Generator().manual_seed(7))
    m = DINO_RESNET(return_layers=[1, 2, 3, 4], return_multilayer=True, add_norm=False, fixed_size=480, weights=sd).to(dev)
    out = m(images.to(dev))
    ref = ores.resnet_dense_features(sd, images, [1, 2, 3, 4], fixed_size=480, add_norm=False)
    assert [tuple(o.shape[1:]) for o in out] == [(256, 120, 120), (512, 60, 60), (1024, 30, 30), (2048, 15, 15)]
    for j, (o, r) in enumerate(zip(out, ref)):
        assert tuple(o.shape) == tuple(r.shape)
        assert rel_l2(o.cpu().numpy(), r.numpy()) < 1e-3, j


# ------------------------------------------------------------------------------------------------ full-size properties (#3, #4, #5)
def _two_runs(make_step):
    outs = []
    for _ in range(2):
        outs.append(make_step())
    return outs


def test_config3_full_size_properties(dev):
    """#3 at its real size (480^2, hidden 512, B=2): finite loss, bit-reproducible across two identically seeded runs."""
    from evals.models.mocov3_res50 import MoCoV3_RES
    from evals.models.probes import SurfaceNormalHead
    from mvp.optim import FlatAdamW
    from mvp.train import train_snorm_step

    g = torch.Generator().manual_seed(0)
    B = 2
    img = torch.randn(B, 3, 480, 480, generator=g).to(dev)
    n = torch.randn(B, 3, 480, 480, generator=g)
    tgt = (n / n.norm(dim=1, keepdim=True)).to(dev)
    mask = (torch.rand(B, 1, 480, 480, generator=g) > 0.1).to(dev)

    def run():
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = MoCoV3_RES(return_layers=[1, 2, 3, 4], return_multilayer=True, add_norm=True, init_seed=3).to(dev)
        torch.manual_seed(1)
        probe = SurfaceNormalHead(feat_dim=m.feat_dim, head_type="dpt", uncertainty_aware=True, hidden_dim=512, kernel_size=3).to(dev)
        opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}])
        losses = [train_snorm_step(m, probe, opt, None, img, tgt, mask).item() for _ in range(2)]
        return losses, opt.flat_param.clone()

    (l0, p0), (l1, p1) = _two_runs(run)
    assert all(np.isfinite(l0)) and l0 == l1 and torch.equal(p0, p1)
    assert torch.isfinite(p0).all()


def test_config4_full_size_properties(dev):
    """#4 at its real size (MAE ViT-B/16 @512^2 -> 32x32 tokens, DPT hidden 512, B=2): finite, reproducible, depth inside the bin range."""
    from evals.models.mae import MAE
    from evals.models.probes import DepthHead
    from evals.utils.losses import DepthLoss
    from mvp.optim import FlatAdamW
    from mvp.train import train_depth_step

    g = torch.Generator().manual_seed(0)
    B = 2
    img = torch.randn(B, 3, 512, 512, generator=g).to(dev)
    tgt = (torch.rand(B, 1, 512, 512, generator=g) * 9.9 + 0.05).to(dev)

    def run():
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = MAE(return_multilayer=True, add_norm=True, init_seed=4).to(dev)
        m.resize_pos_embed(image_size=(512, 512))
        torch.manual_seed(1)
        probe = DepthHead(feat_dim=m.feat_dim, head_type="dpt", kernel_size=3, prediction_type="bindepth", hidden_dim=512).to(dev)
        opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}])
        losses = [train_depth_step(m, probe, opt, None, DepthLoss(), img, tgt.clone()).item() for _ in range(2)]
        with torch.no_grad():
            d = probe(m(img))
        return losses, opt.flat_param.clone(), d

    (l0, p0, d0), (l1, p1, d1) = _two_runs(run)
    assert all(np.isfinite(l0)) and l0 == l1 and torch.equal(p0, p1) and torch.equal(d0, d1)
    assert tuple(d0.shape) == (B, 1, 512, 512) and float(d0.min()) >= 0.001 and float(d0.max()) <= 10.0


def test_config5_full_size_properties(dev):
    """#5 at its real size (iBOT ViT-B/16 @800^2, N=2501, one pair, 20 keypoints): index ranges, finite values, reproducible, and
    correspondences of an image with itself land on the keypoints' own cells."""
    from evals.models.ibot import iBOT
    from mvp import spair

    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = iBOT(output="dense", layer=-1).to(dev)
    g = torch.Generator().manual_seed(0)
    imgs = torch.randn(2, 3, 800, 800, generator=g).to(dev)
    kp = torch.rand(20, 2, generator=g)
    with torch.no_grad():
        f = m(imgs)
        xy, val = spair.correspondence(f[0], f[1], kp)
        xy2, val2 = spair.correspondence(f[0], f[1], kp)
    assert tuple(f.shape) == (2, 768, 50, 50)
    assert int(xy.min()) >= 0 and int(xy.max()) < 50 and torch.isfinite(val).all()
    assert torch.equal(xy, xy2) and torch.equal(val, val2)
    with torch.no_grad():
        cells = (torch.arange(10, dtype=torch.float32) * 5 + 2) / 49.0   # exactly on cell centres of the 50x50 map
        kp_c = torch.stack((cells, cells.flip(0)), 1)
        xy_s, val_s = spair.correspondence(f[0], f[0], kp_c)
    np.testing.assert_array_equal(xy_s.cpu().numpy(), torch.stack(((cells * 49).round().long(), (cells.flip(0) * 49).round().long()), 1).numpy())
    assert float(val_s.min()) > 0.999


def test_config5_full_size_features_and_correspondences_vs_oracle(dev):
    """#5 at its real size against the golden-pinned oracle, not only through properties: iBOT ViT-B/16 (dense, last block) on an
    800x800 pair — N = 2501 tokens per image, the streaming attention kernel at the longest sequence of BASELINE's configs — within
    1e-3 rel-L2 of the oracle's features per image (north_star's tolerance on fp32 features), and the 20-keypoint correspondences
    (evaluate_spair_correspondence.py:45-103) computed by the product from ITS features equal to the oracle's from its own wherever
    the oracle's best cell leads its runner-up by more than the feature error can move a cosine similarity."""
    from evals.models.ibot import iBOT
    from mvp import spair
    from oracle import spair as ospair, vit as ovit

    vsd = ovit.make_vit_weights(seed=31)
    g = torch.Generator().manual_seed(32)
    imgs = torch.randn(2, 3, 800, 800, generator=g)
    kps = torch.rand(20, 2, generator=g)
    with torch.no_grad():
        ref = ovit.vit_dense_features(vsd, imgs, [11], heads=12, add_norm=False)
        m = iBOT(output="dense", layer=-1, weights=vsd).to(dev)
        f = m(imgs.to(dev))
        xy, val = spair.correspondence(f[0], f[1], kps)
    assert tuple(ref.shape) == tuple(f.shape) == (2, 768, 50, 50)
    errs = [rel_l2(f[i].cpu().numpy(), ref[i].numpy()) for i in range(2)]
    print(f"iBOT 800x800 (N=2501) feature rel-L2 per image: {errs[0]:.2e} {errs[1]:.2e}")
    assert max(errs) < 1e-3
    pred, heat = ospair.correspondence(ref, kps)
    top2 = heat.flatten(1).topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-3  # cosine similarities: a 1e-3 feature error moves them by less than that
    assert int(clear.sum()) >= 10
    np.testing.assert_array_equal(xy.cpu().numpy()[clear.numpy()], pred.numpy()[clear.numpy()])
    assert rel_l2(val.cpu().numpy(), top2[:, 0].numpy()) < 1e-3
