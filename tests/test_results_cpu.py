"""N2 second half: experiment name + result-CSV row (train_depth.py:582-601, 676-831)."""
import csv
from types import SimpleNamespace

import torch


def _fixtures():
    cfg = {"optimizer": {"n_epochs": 10, "warmup_epochs": 1.5, "probe_lr": 0.0005, "model_lr": 0.0}, "system": {"random_seed": 8, "num_gpus": 2},
           "batch_size": 16, "note": ""}
    model = SimpleNamespace(checkpoint_name="dino_vitb16", patch_size=16, layer="2-5-8-11", output="dense")
    probe = SimpleNamespace(name="bindepth_dpt_k3")
    keys = ["d1", "d2", "d3", "rmse", "mean_pred", "std_pred", "variance_pred", "mean_gt", "std_gt", "variance_gt", "variance_ratio"]
    sa = {k: torch.tensor(0.5 + i) for i, k in enumerate(keys)}
    from mvp.results import STUFF_THINGS
    sa.update({k: torch.tensor(0.25) for k in STUFF_THINGS})
    si = {k: torch.tensor(1.5 + i) for i, k in enumerate(keys)}  # no stuff/things -> "N/A" columns, as .get(.., 'N/A') does
    lv = {f"level_{i}": {m: torch.tensor(0.1 * i) for m in ("d1", "d2", "d3", "rmse")} for i in range(1, 6)}
    return cfg, model, probe, sa, si, lv


def test_experiment_name_and_row_layout(tmp_path):
    from mvp import results as R

    cfg, model, probe, sa, si, lv = _fixtures()
    ts, name, info = R.experiment_info(cfg, model, probe, "nyuv2", "nyuv2", timestamp="04102026-1200")
    assert name == "04102026-1200_dino_vitb16_16_2-5-8-11_dense_bindepth_dpt_k3_8_10_1.50_0.0005_0.0_32_nyuv2_nyuv2"
    assert len(info) == 13 and info[0] == f"{'dino_vitb16':40s}" and info[10].strip() == "32"
    titles, row = R.depth_result_row(ts, info, sa, si, lv, lv, "/x/ckpt.pth", "nyuv2")
    assert len(titles) == len(row) == 14 + 21 + 11 + 20 + 20 + 20 + 1
    assert titles[:3] == ["Timestamp", "Model Checkpoint", "Patch Size"] and titles[14] == "d1 SA" and titles[14 + 21] == "d1 SI"
    assert titles[14 + 32] == "Level level_1 d1 SA" and titles[-1] == "ckpt_path" and titles[-11] == "stuff_d1 SI" and titles[-2] == "things_pixels SI"
    assert row[14 + 32] == "0.1000" and row[-11] == "N/A" and row[-21] == "0.25" and row[-1] == "/x/ckpt.pth"
    t2, r2 = R.depth_result_row(ts, info, sa, si, lv, lv, "/x/ckpt.pth", "navi_reldepth")
    assert len(t2) == len(r2) == len(titles) - 20
    path = R.result_csv_path(str(tmp_path), "depth", "nyuv2", add_norm=True)
    assert path.endswith("result/depth/depth_results_nyuv2_final_with_batchnorm.csv")
    R.append_result_csv(path, titles, row)
    R.append_result_csv(path, titles, row)
    rows = list(csv.reader(open(path)))
    assert len(rows) == 3 and rows[0] == titles and rows[1] == rows[2] == row
