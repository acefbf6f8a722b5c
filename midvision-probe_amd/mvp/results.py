"""Experiment naming and the result-CSV row of the trainers (reference: train_depth.py:582-601 for ``exp_name``,
train_depth.py:676-829 for the row layout, the column titles and the append-to-csv protocol), so that downstream analysis
scripts that read ``<output_dir>/result/depth/depth_results_<test_dset>_final[_with_batchnorm].csv`` keep working.

Layout of a row: timestamp | model: checkpoint_name, patch_size, layer, output | probe name | seed, n_epochs, warmup_epochs,
probe_lr, model_lr, global batch, train dataset, test dataset | every scale-aware global metric | every scale-invariant global
metric | "Level k metric" SA then SI (4 decimals) | (unless dataset == navi_reldepth) stuff/things x 10 SA then SI | ckpt path."""
from __future__ import annotations

import csv
import os
from datetime import datetime
from typing import Dict, List, Optional, Sequence, Tuple

STUFF_THINGS = ["stuff_d1", "stuff_d2", "stuff_d3", "stuff_rmse", "stuff_pixels", "things_d1", "things_d2", "things_d3", "things_rmse", "things_pixels"]
_FIXED_TITLES = ["Timestamp", "Model Checkpoint", "Patch Size", "Layer", "Model Output", "Probe Name", "Random Seed", "Num Epochs", "Warmup Epochs",
                 "Probe LR", "Model LR", "Batch Size", "Train Dataset", "Test Dataset"]


def experiment_info(cfg, model, probe, train_dset: str, test_dset: str, timestamp: Optional[str] = None) -> Tuple[str, str, List[str]]:
    """-> (timestamp, exp_name, exp_info fields).  Field formatting as train_depth.py:582-601 (fixed widths, spaces stripped from the name)."""
    timestamp = timestamp or datetime.now().strftime("%d%m%Y-%H%M")
    opt, sys_ = cfg["optimizer"], cfg["system"]
    model_info = [f"{model.checkpoint_name:40s}", f"{model.patch_size:2d}", f"{str(model.layer):5s}", f"{model.output:10s}"]
    probe_info = [f"{probe.name:25s}"]
    batch = cfg["batch_size"] * sys_["num_gpus"]
    train_info = [f"{sys_['random_seed']}", f"{opt['n_epochs']:3d}", f"{opt['warmup_epochs']:4.2f}", f"{str(opt['probe_lr']):>10s}",
                  f"{str(opt['model_lr']):>10s}", f"{batch:4d}", f"{train_dset:10s}", f"{test_dset:10s}"]
    exp_name = "_".join([timestamp] + model_info + probe_info + train_info)
    note = cfg.get("note", "")
    exp_name = (f"{exp_name}_{note}" if note != "" else exp_name).replace(" ", "")
    info = [s.replace(",", "-") for s in model_info + probe_info + train_info]
    return timestamp, exp_name, info


def depth_result_row(timestamp: str, exp_info: Sequence[str], sa_global: Dict, si_global: Dict, sa_levels: Dict, si_levels: Dict,
                     ckpt_path: str, dataset_name: str) -> Tuple[List[str], List[str]]:
    """-> (column_titles, row) exactly as train_depth.py:676-805 builds them."""
    row = [f"{sa_global.get(m, 'N/A')}" for m in sa_global] + [f"{si_global.get(m, 'N/A')}" for m in si_global]
    for levels in (sa_levels, si_levels):
        for lvl in levels:
            for m in levels[lvl]:
                row.append(f"{levels[lvl][m]:.4f}")
    titles = list(_FIXED_TITLES) + [f"{m} SA" for m in sa_global] + [f"{m} SI" for m in si_global]
    titles += [f"Level {lvl} {m} SA" for lvl in sa_levels for m in sa_levels[lvl]] + [f"Level {lvl} {m} SI" for lvl in si_levels for m in si_levels[lvl]]
    if dataset_name != "navi_reldepth":
        row += [f"{sa_global.get(m, 'N/A')}" for m in STUFF_THINGS] + [f"{si_global.get(m, 'N/A')}" for m in STUFF_THINGS]
        titles += [f"{m} SA" for m in STUFF_THINGS] + [f"{m} SI" for m in STUFF_THINGS]
    titles.append("ckpt_path")
    return titles, [timestamp] + list(exp_info) + row + [str(ckpt_path)]


def result_csv_path(output_dir: str, task: str, test_dset: str, add_norm: bool) -> str:
    """train_depth.py:808-819: <output_dir>/result/<task>/<task>_results_<test_dset>_final[_with_batchnorm].csv"""
    name = f"{task}_results_{test_dset}_final_with_batchnorm.csv" if add_norm else f"{task}_results_{test_dset}_final.csv"
    return os.path.join(f"{output_dir}/result", task, name)


def append_result_csv(path: str, titles: Sequence[str], row: Sequence[str]) -> str:
    """Header once (new or empty file), then one appended row per run (train_depth.py:823-831)."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    if not os.path.exists(path) or os.stat(path).st_size == 0:
        with open(path, "a", newline="") as f:
            csv.writer(f).writerow(titles)
    with open(path, "a", newline="") as f:
        csv.writer(f).writerow(row)
    return path
