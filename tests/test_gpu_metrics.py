"""GPU parity of the validation metrics against the REFERENCE's own outputs (golden metrics.npz)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_depth_metrics_vs_reference(dev):
    from evals.utils.metrics import evaluate_depth, match_scale_and_shift

    g = load_golden("metrics.npz")
    pr, gt = torch.from_numpy(g["pred"]).to(dev), torch.from_numpy(g["gt"]).to(dev)
    np.testing.assert_allclose(match_scale_and_shift(pr, gt).cpu().numpy(), g["matched"], rtol=5e-5, atol=5e-5)
    for tag, si in (("sa", False), ("si", True)):
        gm, lv, seg = evaluate_depth(pr, gt, None, scale_invariant=si, is_navi=True)
        assert sorted(lv) == [f"level_{i}" for i in range(1, 6)] and seg == []
        for k, v in gm.items():
            tol = 2e-3 if (tag == "si" and "variance" in k or tag == "si" and k == "std_pred") else 1e-4
            np.testing.assert_allclose(v.reshape(-1).numpy(), g[f"{tag}_{k}"], rtol=tol, atol=2e-6, err_msg=f"{tag}_{k}")
    avg = evaluate_depth(pr, gt, None, image_average=True, is_navi=True)[0]
    assert abs(avg["rmse"].item() - g["sa_rmse"].mean()) < 1e-4


def test_snorm_metrics_vs_reference(dev):
    from evals.utils.metrics import evaluate_surface_norm

    g = load_golden("metrics.npz")
    gm, _, _ = evaluate_surface_norm(torch.from_numpy(g["sn_pred"]).to(dev), torch.from_numpy(g["sn_gt"]).to(dev), None, is_navi=True)
    for k, v in gm.items():
        np.testing.assert_allclose(v.numpy(), g[f"sn_{k}"], rtol=5e-5, atol=1e-6, err_msg=k)


def test_validate_loop_depth_rmse_vs_oracle(dev):
    """validate() (train_depth.py:357-483): loss + global metrics over a 2-batch loader; depth RMSE of the
    HIP path vs the CPU oracle within the north-star 1e-2."""
    from evals.models.dino import DINO
    from evals.models.probes import DepthHead
    from evals.utils.losses import DepthLoss
    from mvp.train import validate
    from oracle import metrics as om, probes as oprobes, train as otrain, vit as ovit

    D = 128
    vsd = ovit.make_vit_weights(embed_dim=D, depth=4, seed=61)
    psd = oprobes.make_linear_head_weights([D] * 4, 256, 1, seed=62)
    model = DINO(return_multilayer=True, add_norm=True, weights=vsd).to(dev).eval()
    probe = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type="bindepth")
    probe.load_state_dict(psd, strict=True)
    probe = probe.to(dev).eval()
    batches = []
    for s in range(2):
        im, d = otrain.synthetic_depth_batch(3, 64, 80, rank=0, step=s)
        batches.append({"image": im, "depth": d})
    loss, metrics, levels = validate(model, probe, batches, DepthLoss(), is_navi=True)
    assert sorted(levels) == [f"level_{i}" for i in range(1, 6)]
    # oracle (eval mode: running stats = init (0,1))
    tr = otrain.DepthProbeTrainer(vsd, psd, layers=(0, 1, 2, 3), heads=2)
    rm = []
    for b in batches:
        with torch.no_grad():
            f = ovit.vit_dense_features(vsd, b["image"], [0, 1, 2, 3], heads=2, bn_running=tr.bn_running, bn_training=False)
            _, pred = tr.forward_loss(f, b["depth"].clone())
        rm.append(om.depth_global_metrics(pred.detach(), b["depth"])["rmse"])
    ref_rmse = torch.cat(rm).mean().item()
    assert abs(metrics["rmse"] - ref_rmse) < 1e-2 * ref_rmse
    assert 0.0 <= metrics["d1"] <= 1.0 and loss > 0


def _check(g, prefix, gm, lv, sm, rtol=1e-4):
    for k in [k for k in gm if k.startswith(("stuff_", "things_"))]:
        np.testing.assert_allclose(gm[k].reshape(-1).numpy(), g[f"{prefix}_{k}"], rtol=rtol, atol=2e-6, err_msg=f"{prefix}_{k}")
    assert len([k for k in gm if k.startswith(("stuff_", "things_"))]) == 10
    for L, d in lv.items():
        for k, v in d.items():
            np.testing.assert_allclose(v.numpy(), g[f"{prefix}_{L}_{k}"], rtol=rtol, atol=2e-6, err_msg=f"{prefix}_{L}_{k}")
    ref = g[f"{prefix}_segments"]
    got = np.array([[m["segment_id"], m["image_idx"], m["area"], m["d1_ratio"]] for m in sm], dtype=np.float64)
    assert got.shape == ref.shape
    np.testing.assert_array_equal(got[:, :2], ref[:, :2])          # ids (torch.unique order) x image index: exact
    np.testing.assert_allclose(got[:, 2], ref[:, 2], rtol=0, atol=0)  # areas are pixel counts: exact
    np.testing.assert_allclose(got[:, 3], ref[:, 3], rtol=rtol, atol=2e-6)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_depth_breakdown_vs_reference(dev, tag):
    """metrics.py:179-358: stuff/things, 5 centroid levels and per-segment d1 from ONE segmented reduction, held to the
    outputs of the reference's own evaluate_depth (golden metrics_seg.npz), scale-aware and scale-invariant."""
    from evals.utils.metrics import evaluate_depth

    g = load_golden("metrics_seg.npz")
    pr, gt = torch.from_numpy(g[f"{tag}_pred"]).to(dev), torch.from_numpy(g[f"{tag}_gt"]).to(dev)
    seg = torch.from_numpy(g[f"{tag}_seg"]).long().to(dev)
    for mode, si in (("sa", False), ("si", True)):
        gm, lv, sm = evaluate_depth(pr, gt, seg, scale_invariant=si, is_navi=False)
        _check(g, f"{tag}_{mode}", gm, lv, sm, rtol=1e-4 if not si else 5e-4)
    gm, lv, _ = evaluate_depth(pr, gt, seg, image_average=True, num_levels=3, is_navi=False)
    assert abs(gm["rmse"].item() - float(g[f"{tag}_avg3_rmse"])) < 1e-4
    for L, d in lv.items():
        assert abs(d["d1"].item() - float(g[f"{tag}_avg3_{L}_d1"])) < 1e-5
    with pytest.raises(ValueError):
        evaluate_depth(pr, gt, None, is_navi=False)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_snorm_breakdown_vs_reference(dev, tag):
    from evals.utils.metrics import evaluate_surface_norm

    g = load_golden("metrics_seg.npz")
    pr, gt = torch.from_numpy(g[f"{tag}_sn_pred"]).to(dev), torch.from_numpy(g[f"{tag}_sn_gt"]).to(dev)
    seg = torch.from_numpy(g[f"{tag}_seg"]).long().to(dev)
    gm, lv, sm = evaluate_surface_norm(pr, gt, seg, is_navi=False)
    _check(g, f"{tag}_sn", gm, lv, sm, rtol=2e-4)


def test_breakdown_full_size_bins_partition_the_image(dev):
    """480x640, B=4, 150 ids: level bins and segment bins each count every valid pixel exactly once; counts are integers."""
    from evals.utils import metrics as M

    g = torch.Generator().manual_seed(3)
    B, H, W = 4, 480, 640
    pr = (torch.rand(B, H, W, generator=g) * 9 + 0.05).to(dev)
    gt = torch.rand(B, H, W, generator=g) * 9 + 0.05
    gt[torch.rand(B, H, W, generator=g) < 0.15] = 0
    seg = torch.randint(0, 150, (B, H, W), generator=g)
    lv, sg = M._breakdown(pr, gt.to(dev), seg.to(dev), None, 0, 5)
    nvalid = (gt > 0).sum(dim=(1, 2)).double()
    assert torch.equal(lv[:, :, 0].sum(1), nvalid) and torch.equal(sg[:, :, 1].sum(1), nvalid)
    assert torch.equal(sg[:, :, 0].sum(1), torch.full((B,), float(H * W), dtype=torch.float64))
    assert torch.equal(lv[:, :, 1:4], lv[:, :, 1:4].round()) and torch.equal(lv[:, :, 1:4].sum(1), sg[:, :, 2:5].sum(1))
    np.testing.assert_allclose(lv[:, :, 4].sum(1).numpy(), sg[:, :, 5].sum(1).numpy(), rtol=1e-12)
    lv2, sg2 = M._breakdown(pr, gt.to(dev), seg.to(dev), None, 0, 5)
    assert torch.equal(lv[:, :, :4], lv2[:, :, :4]) and torch.equal(sg[:, :, :5], sg2[:, :, :5])
