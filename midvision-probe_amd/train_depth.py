#!/usr/bin/env python3
"""Entry point mirroring the reference's train_depth.py (hydra.main -> train_model, train_depth.py:542-855)
for the hot path on MI355X: compose config -> instantiate backbone & probe -> FlatAdamW + LambdaLR ->
train() -> validate() -> ckpt.pth.  Data: evals.datasets.build_loader (reference sampler semantics) over the
synthetic NYU-shaped dataset, fed through the pinned double-buffered H2D prefetcher (the reference's dataset
decoders, W&B, CSV bookkeeping are out of scope).

    python train_depth.py backbone=dino_b16 +backbone.return_multilayer=True probe=depth_linear batch_size=16
    torchrun --nproc-per-node N --master-addr 127.0.0.1 train_depth.py ...      # one rank per GPU
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402

from mvp import checkpoint, config  # noqa: E402
from mvp import dist as mdist  # noqa: E402
from mvp.optim import FlatAdamW  # noqa: E402
from mvp.train import train, validate  # noqa: E402


def main(argv):
    from evals.utils.losses import DepthLoss
    from evals.utils.optim import cosine_decay_linear_warmup

    cfg = config.compose("depth_training", argv)
    rank, local, world = mdist.env_setup("nccl")
    dev = torch.device("cuda", torch.cuda.current_device())
    from evals.datasets import build_loader

    ds = cfg["dataset"]
    loader = build_loader(dict(ds, batch_size=cfg["batch_size"]), "train", cfg["batch_size"], num_gpus=world, num_workers=cfg.get("num_workers", 2),
                          with_snorm=False)
    model = config.instantiate(cfg["backbone"]).to(dev)
    probe = config.instantiate(cfg["probe"], feat_dim=model.feat_dim, max_depth=ds["max_depth"]).to(dev)
    opt = FlatAdamW([{"params": probe.parameters(), "lr": cfg["optimizer"]["probe_lr"]}])
    n_ep = cfg["optimizer"]["n_epochs"]
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda e: cosine_decay_linear_warmup(
        e, n_ep * len(loader), cfg["optimizer"]["warmup_epochs"] * len(loader)))
    hist = train(model, probe, loader, opt, sched, n_ep, detach_model=True, loss_fn=DepthLoss(), rank=rank, world_size=world)
    if rank == 0:
        model.eval(); probe.eval()
        vloader = build_loader(dict(ds, num_batches=2, batch_size=cfg["batch_size"]), "valid", cfg["batch_size"], with_snorm=False)
        vloss, metrics = validate(model, probe, vloader, DepthLoss())
        print(f"train loss/epoch {hist}  | valid loss {vloss:.4f} " + " ".join(f"{k} {v:.4f}" for k, v in metrics.items() if k in ("d1", "rmse")))
        out = os.path.join(cfg["output_dir"], "depth_exps", f"{model.checkpoint_name}_{probe.name}".replace("$", ""))
        print("saved", checkpoint.save_checkpoint(os.path.join(out, "ckpt.pth"), cfg, model, probe))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
