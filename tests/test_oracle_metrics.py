"""Pin oracle/metrics.py to the reference's evals/utils/metrics.py outputs (golden metrics.npz)."""
import numpy as np
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
