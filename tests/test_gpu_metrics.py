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
        assert lv == {} and seg == []
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
    loss, metrics = validate(model, probe, batches, DepthLoss())
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
