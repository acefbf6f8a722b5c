#!/usr/bin/env python3
"""Entry point mirroring the reference's train_depth.py (hydra.main -> train_model, train_depth.py:542-855)
for the hot path on MI355X: compose config -> instantiate backbone & probe -> FlatAdamW + LambdaLR ->
train() -> validate() -> ckpt.pth.  Data: synthetic NYU-shaped batches (the reference's dataset loaders,
W&B, CSV bookkeeping are out of scope).

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


class SyntheticDepth:
    """Batch dict contract of evals/datasets/nyu.py:245-251: {"image": [B,3,H,W], "depth": [B,1,H,W]}."""

    def __init__(self, n, B, hw, rank, max_depth=10):
        self.n, self.B, self.hw, self.rank, self.max_depth, self.name = n, B, tuple(hw), rank, max_depth, "synthetic"

    def __len__(self):
        return self.n

    def __iter__(self):
        for s in range(self.n):
            g = torch.Generator().manual_seed(1000 * self.rank + s)
            img = torch.randn(self.B, 3, *self.hw, generator=g)
            d = torch.rand(self.B, 1, *self.hw, generator=g) * 9.9 + 0.05
            d[torch.rand(self.B, 1, *self.hw, generator=g) < 0.1] = 0
            yield {"image": img, "depth": d}


def main(argv):
    from evals.utils.losses import DepthLoss
    from evals.utils.optim import cosine_decay_linear_warmup

    cfg = config.compose("depth_training", argv)
    rank, local, world = mdist.env_setup("nccl")
    dev = torch.device("cuda", torch.cuda.current_device())
    ds = cfg["dataset"]
    loader = SyntheticDepth(ds["num_batches"], cfg["batch_size"], ds["image_size"], rank, ds["max_depth"])
    model = config.instantiate(cfg["backbone"]).to(dev)
    probe = config.instantiate(cfg["probe"], feat_dim=model.feat_dim, max_depth=ds["max_depth"]).to(dev)
    opt = FlatAdamW([{"params": probe.parameters(), "lr": cfg["optimizer"]["probe_lr"]}])
    n_ep = cfg["optimizer"]["n_epochs"]
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda e: cosine_decay_linear_warmup(
        e, n_ep * len(loader), cfg["optimizer"]["warmup_epochs"] * len(loader)))
    hist = train(model, probe, loader, opt, sched, n_ep, detach_model=True, loss_fn=DepthLoss(), rank=rank, world_size=world)
    if rank == 0:
        model.eval(); probe.eval()
        vloss, metrics = validate(model, probe, SyntheticDepth(2, cfg["batch_size"], ds["image_size"], 99), DepthLoss())
        print(f"train loss/epoch {hist}  | valid loss {vloss:.4f} " + " ".join(f"{k} {v:.4f}" for k, v in metrics.items() if k in ("d1", "rmse")))
        out = os.path.join(cfg["output_dir"], "depth_exps", f"{model.checkpoint_name}_{probe.name}".replace("$", ""))
        print("saved", checkpoint.save_checkpoint(os.path.join(out, "ckpt.pth"), cfg, model, probe))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
