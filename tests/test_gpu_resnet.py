"""GPU parity of the ResNet-50 backbone wrappers against the CPU oracle.
The trunk arithmetic (torchvision 0.17.1) is absent from /root/reference: "parity unpinned" for it;
the wrapper logic (stages, taps, per-stage BatchNorm2d indexing, Resize) is restated from the
reference source text in oracle/resnet.py."""
import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rand_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    aff = []
    with torch.no_grad():
        for bn in model.batchnorms:
            bn.weight.copy_(1 + 0.2 * torch.randn(bn.weight.shape, generator=g))
            bn.bias.copy_(0.1 * torch.randn(bn.bias.shape, generator=g))
            aff.append((bn.weight.detach().cpu().clone(), bn.bias.detach().cpu().clone()))
    return aff


def test_dino_resnet50_multilayer(dev):
    from evals.models.dino_res50 import DINO_RESNET
    from oracle import resnet as ores

    sd = ores.make_resnet50_weights(seed=3)
    images = torch.randn(2, 3, 96, 96, generator=torch.Generator().manual_seed(1))
    m = DINO_RESNET(return_layers=[1, 2, 3, 4], return_multilayer=True, add_norm=True, fixed_size=96, weights=sd).to(dev)
    assert m.feat_dim == [(256, 120), (512, 60), (1024, 30), (2048, 15)] and m.layer == "1-2-3-4" and m.checkpoint_name == "dino_resnet50"
    aff = _rand_bn(m, 2)
    out = m(images.to(dev))
    run = [(torch.zeros(c), torch.ones(c)) for c, _ in m.feat_dims]
    ref = ores.resnet_dense_features(sd, images, [1, 2, 3, 4], fixed_size=96, bn_affine=aff, bn_running=run)
    for j, (o, r) in enumerate(zip(out, ref)):
        assert tuple(o.shape) == tuple(r.shape)
        assert rel_l2(o.cpu().numpy(), r.numpy()) < 1e-3, j
    for i in (1, 2, 3, 4):
        assert rel_l2(m.batchnorms[i].running_var.cpu().numpy(), run[i][1].numpy()) < 1e-3
    assert int(m.batchnorms[0].num_batches_tracked) == 0 and int(m.batchnorms[4].num_batches_tracked) == 1


def test_mocov3_resnet50_single_layer_with_upsample(dev):
    """BASELINE config #1 shape at test size: single last-stage tap, input smaller than fixed_size (bilinear upsample)."""
    from evals.models.mocov3_res50 import MoCoV3_RES
    from oracle import resnet as ores

    sd = ores.make_resnet50_weights(seed=4)
    images = torch.randn(2, 3, 64, 80, generator=torch.Generator().manual_seed(5))
    m = MoCoV3_RES(return_layers=[1, 2, 3, 4], output="dense-cls", add_norm=True, fixed_size=128, weights=sd).to(dev)
    assert m.feat_dim == (2048, 15) and m.multilayers == [4]
    out = m(images.to(dev))
    ref = ores.resnet_dense_features(sd, images, [4], fixed_size=128)
    assert tuple(out.shape) == (2, 2048, 4, 4)
    assert rel_l2(out.cpu().numpy(), ref.numpy()) < 1e-3
    raw = MoCoV3_RES(return_layers=[1, 2, 3, 4], add_norm=False, fixed_size=128, weights=sd).to(dev)(images.to(dev))
    assert rel_l2(raw.cpu().numpy(), ores.resnet_dense_features(sd, images, [4], fixed_size=128, add_norm=False).numpy()) < 1e-3


@pytest.mark.parametrize("shape", [(2, 96, 96), (1, 61, 83), (3, 480, 480)])
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_fused_stem_vs_torch(dev, shape, precision):
    """csrc/stem.hip: conv 7x7/2 (folded BN) + ReLU + max-pool 3x3/2 in one kernel vs fp64 torch, odd sizes (ragged tiles, image
    borders), and vs the older im2col + GEMM + pool form."""
    import torch.nn.functional as F
    from mvp import lib, ops
    from mvp.vit import parse_precision

    B, H, W = shape
    pr = parse_precision(precision)
    g = torch.Generator().manual_seed(H * W + B)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    b = torch.randn(64, generator=g) * 0.1
    wk = torch.zeros(64, 160)
    wk[:, :147] = w.permute(0, 2, 3, 1).reshape(64, -1)
    wp = ops.split_bf16(wk.to(dev), pr)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
    of = torch.full((B * Hp * Wp, 64), float("nan"), device=dev)
    op = ops.empty_pair((B * Hp * Wp, 64), pr, dev)
    lib.call("mvp_stem7x7_pool", lib.StemArgs(lib.ptr(x.to(dev)), lib.ptr(wp[0]), lib.ptr(wp[1]), lib.ptr(b.to(dev)), lib.ptr(of), lib.ptr(op[0]), lib.ptr(op[1]),
                                               B, H, W, pr))
    torch.cuda.synchronize()
    xr, wr = (x, w) if pr == lib.PREC_BF16X3 else (x.bfloat16().float(), w.bfloat16().float())
    ref = F.max_pool2d(F.relu(F.conv2d(xr.double(), wr.double(), b.double(), stride=2, padding=3)), 3, 2, 1)
    assert tuple(ref.shape) == (B, 64, Hp, Wp)
    got = of.view(B, Hp, Wp, 64).permute(0, 3, 1, 2).cpu()
    tol = 2e-5 if pr == lib.PREC_BF16X3 else 1e-5
    assert rel_l2(got.numpy(), ref.numpy()) < tol
    pair = (op[0].float() + (op[1].float() if op[1] is not None else 0)).view(B, Hp, Wp, 64).permute(0, 3, 1, 2).cpu()
    assert rel_l2(pair.numpy(), ref.numpy()) < (tol + 2e-5 if pr == lib.PREC_BF16X3 else 5e-3)


def test_fused_stem_matches_im2col_path(dev, monkeypatch):
    from mvp.resnet import ResNetEngine
    from oracle import resnet as ores

    sd = ores.make_resnet50_weights(seed=6)
    eng = ResNetEngine(sd, device=dev)
    images = torch.randn(2, 3, 128, 160, generator=torch.Generator().manual_seed(3)).to(dev)
    f1, p1, H1, W1 = eng._stem(images, want_f32=True)
    monkeypatch.setenv("MVP_STEM", "im2col")
    f2, p2, H2, W2 = eng._stem(images)
    assert (H1, W1) == (H2, W2) == (32, 40)
    assert rel_l2(f1.cpu().numpy(), f2.cpu().numpy()) < 1e-5
