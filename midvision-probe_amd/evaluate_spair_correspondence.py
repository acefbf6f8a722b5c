#!/usr/bin/env python3
"""Entry point mirroring the reference's evaluate_spair_correspondence.py main (lines 123-215) for BASELINE config #5 on MI355X:
instantiate the backbone (output="dense"), run compute_errors over image pairs (sharded across ranks when launched under torchrun),
report Recall@0.10.  Data: SPair-shaped synthetic pairs (mvp.spair.SyntheticSPair; the SPair-71k reader is out of scope).

    python evaluate_spair_correspondence.py backbone=ibot_b16 image_size=800 num_instances=16
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402

from mvp import config, spair  # noqa: E402
from mvp import dist as mdist  # noqa: E402


def main(argv):
    cfg = config.compose("spair_correspondence", argv)
    rank, local, world = mdist.env_setup("nccl")
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(int(cfg["random_seed"]))
    model = config.instantiate(cfg["backbone"], output="dense", return_multilayer=cfg["multilayer"]).to(dev)
    ds = spair.SyntheticSPair(num_pairs=int(cfg["num_instances"]), image_size=int(cfg["image_size"]), seed=int(cfg["random_seed"]))
    recall, confusion = spair.evaluate_dataset(model, ds, 0.10, rank=rank, world=world)
    if rank == 0:
        print(f"Recall@0.10 {model.checkpoint_name} layer {model.layer} ({len(ds)} synthetic pairs, {world} rank(s)) | {recall:6.2f}")
    if world > 1:
        torch.distributed.destroy_process_group()
    return recall


if __name__ == "__main__":
    main(sys.argv[1:])
