"""Pin oracle/metrics.py to the reference's evals/utils/metrics.py outputs (golden metrics.npz)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import metrics as om


def test_depth_metrics_and_scale_shift():
    g = load_golden("metrics.npz")
    pr, gt = torch.from_numpy(g["pred"]), torch.from_numpy(g["gt"])
    np.testing.assert_allclose(om.match_scale_and_shift(pr, gt).numpy(), g["matched"], rtol=2e-5, atol=2e-5)
    for tag, si in (("sa", False), ("si", True)):
        m = om.depth_global_metrics(pr, gt, scale_invariant=si)
        for k, v in m.items():
            np.testing.assert_allclose(v.numpy(), g[f"{tag}_{k}"], rtol=5e-5, atol=1e-6, err_msg=f"{tag}_{k}")


def test_snorm_metrics():
    g = load_golden("metrics.npz")
    m = om.snorm_global_metrics(torch.from_numpy(g["sn_pred"]), torch.from_numpy(g["sn_gt"]))
    for k, v in m.items():
        np.testing.assert_allclose(v.numpy(), g[f"sn_{k}"], rtol=2e-5, atol=1e-6, err_msg=k)


def _check_breakdown(g, tag, prefix, groups, levels, segments, rtol=2e-5):
    for k, v in groups.items():
        np.testing.assert_allclose(v.numpy(), g[f"{prefix}_{k}"], rtol=rtol, atol=1e-6, err_msg=f"{prefix}_{k}")
    for L, d in levels.items():
        for k, v in d.items():
            np.testing.assert_allclose(v.numpy(), g[f"{prefix}_{L}_{k}"], rtol=rtol, atol=1e-6, err_msg=f"{prefix}_{L}_{k}")
    ref = g[f"{prefix}_segments"]
    got = np.array(segments, dtype=np.float64)
    assert got.shape == ref.shape
    np.testing.assert_array_equal(got[:, :2], ref[:, :2])            # (segment id, image idx): torch.unique order
    np.testing.assert_allclose(got[:, 2:], ref[:, 2:], rtol=rtol, atol=1e-6)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_depth_breakdown_vs_reference_golden(golden, tag):
    """metrics.py:179-358 (stuff/things, centroid levels, per-segment d1) vs outputs of the reference's evaluate_depth."""
    from oracle import metrics as om

    g = golden("metrics_seg.npz")
    pr, gt, seg = torch.from_numpy(g[f"{tag}_pred"]), torch.from_numpy(g[f"{tag}_gt"]), torch.from_numpy(g[f"{tag}_seg"]).long()
    for mode, si in (("sa", False), ("si", True)):
        _check_breakdown(g, tag, f"{tag}_{mode}", *om.depth_breakdown(pr, gt, seg, scale_invariant=si))
    _, lv3, _ = om.depth_breakdown(pr, gt, seg, num_levels=3)
    for L, d in lv3.items():
        np.testing.assert_allclose(d["d1"].mean().numpy(), g[f"{tag}_avg3_{L}_d1"], rtol=2e-5)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_snorm_breakdown_vs_reference_golden(golden, tag):
    from oracle import metrics as om

    g = golden("metrics_seg.npz")
    pr, gt, seg = torch.from_numpy(g[f"{tag}_sn_pred"]), torch.from_numpy(g[f"{tag}_sn_gt"]), torch.from_numpy(g[f"{tag}_seg"]).long()
    _check_breakdown(g, tag, f"{tag}_sn", *om.snorm_breakdown(pr, gt, seg))


def test_argmax_2d_vs_reference_golden(golden):
    """correspondence.py:179-190: bit-exact (integer indices), ties included, both max and min."""
    from oracle import spair as osp

    g = golden("spair.npz")
    np.testing.assert_array_equal(osp.argmax_2d(torch.from_numpy(g["heat"])).numpy(), g["pred_max"])
    np.testing.assert_array_equal(osp.argmax_2d(torch.from_numpy(g["heat"]), max_value=False).numpy(), g["pred_min"])
    np.testing.assert_array_equal(osp.argmax_2d(torch.from_numpy(g["tie_heat"])).numpy(), g["tie_max"])
    np.testing.assert_array_equal(osp.argmax_2d(torch.from_numpy(g["tie_heat"]), max_value=False).numpy(), g["tie_min"])
    pred, heat = osp.correspondence(torch.from_numpy(g["feats"]), torch.from_numpy(g["kps01"]))
    np.testing.assert_array_equal(pred.numpy(), g["pred_max"])
    np.testing.assert_allclose(heat.numpy(), g["heat"], rtol=1e-5, atol=1e-6)


def test_scale_invariant_train_branch_vs_reference_golden(golden):
    """train_depth.py:114-118 (match_scale_and_shift with DETACHED scale/shift -> clamp -> DepthLoss): loss and input gradient."""
    from oracle import losses as ol
    from oracle import metrics as om

    g = golden("si_train.npz")
    pred = torch.from_numpy(g["pred"]).requires_grad_(True)
    tgt = torch.from_numpy(g["target"])
    p2 = om.match_scale_and_shift(pred, tgt)
    np.testing.assert_allclose(p2.detach().numpy(), g["matched"], rtol=2e-5, atol=2e-6)
    loss = ol.depth_loss(p2.clamp(min=0.001, max=1.0), tgt.clone())
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    np.testing.assert_allclose(pred.grad.numpy(), g["grad"], rtol=1e-4, atol=1e-7)
