#!/usr/bin/env python3
"""Entry point mirroring the reference's train_snorm.py (train_snorm.py:86-120 loop; bicubic upsample,
uncertainty-aware angular loss, SurfaceNormalHead) on synthetic NYU-shaped batches.

    python train_snorm.py backbone=dino_b16 +backbone.return_multilayer=True probe=snorm_dpt batch_size=8
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402

from mvp import checkpoint, config  # noqa: E402
from mvp import dist as mdist  # noqa: E402
from mvp.optim import FlatAdamW  # noqa: E402
from mvp.pipeline import freeze_gc, pipelined_features  # noqa: E402
from mvp.train import train_snorm_step  # noqa: E402


def main(argv):
    from evals.utils.metrics import evaluate_surface_norm
    from evals.utils.optim import cosine_decay_linear_warmup
    from mvp import functional as MF

    cfg = config.compose("snorm_training", argv)
    if float(cfg["optimizer"].get("model_lr", 0.0)) != 0.0:
        raise NotImplementedError("optimizer.model_lr != 0 (backbone fine-tuning) is outside the frozen-backbone hot path")
    rank, local, world = mdist.env_setup("nccl")
    torch.manual_seed(int(cfg["system"]["random_seed"]))
    dev = torch.device("cuda", torch.cuda.current_device())
    ds = cfg["dataset"]
    from evals.datasets import build_loader
    from mvp.prefetch import DevicePrefetcher

    hw, B = tuple(ds["image_size"]), cfg["batch_size"]
    loader = build_loader(dict(ds, batch_size=B), "train", B, num_gpus=world, num_workers=cfg.get("num_workers", 2))
    nb = len(loader)
    model = config.instantiate(cfg["backbone"]).to(dev)
    probe = config.instantiate(cfg["probe"], feat_dim=model.feat_dim).to(dev)
    opt = FlatAdamW([{"params": probe.parameters(), "lr": cfg["optimizer"]["probe_lr"]}], overlap_comm=world > 1)
    n_ep = cfg["optimizer"]["n_epochs"]
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda e: cosine_decay_linear_warmup(e, n_ep * nb, cfg["optimizer"]["warmup_epochs"] * nb))
    freeze_gc()  # no full-heap collector pause inside the loop
    is_eval = bool(cfg.get("is_eval", False))
    if is_eval:  # train_snorm.py:541-542: an evaluation run loads model AND probe of cfg.ckpt_path
        ck = str(cfg.get("ckpt_path", "") or "").replace("\\$", "$")
        if not ck:
            raise SystemExit("is_eval=True needs ckpt_path=<.../ckpt.pth> (refusing to validate a randomly initialised probe)")
        checkpoint.load_checkpoint(ck, model, probe, load_model=True)
    for ep in range(0 if is_eval else n_ep):
        tot = 0.0
        if world > 1:
            loader.sampler.set_epoch(ep)
        for batch, feats in pipelined_features(model, DevicePrefetcher(loader, dev), probe=probe):  # forward of the next batch already in flight
            target = batch["snorm"]
            mask = batch["depth"] > 0                               # train_snorm.py:95
            tot += train_snorm_step(model, probe, opt, sched, None, target, mask, feats=feats).item()
        if rank == 0:
            print(f"epoch {ep} train loss {tot / nb:.4f}")
    opt.finish_pending()
    if rank == 0:
        model.eval(); probe.eval()
        b = next(iter(build_loader(dict(ds, num_batches=1, batch_size=B), "valid", B)))
        with torch.no_grad():
            pred = MF.interpolate(probe(model(b["image"].to(dev))).contiguous(), size=hw, mode="bicubic")
        gm, _, _ = evaluate_surface_norm(pred, b["snorm"].to(dev), None, image_average=True, is_navi=True)
        print("valid " + " ".join(f"{k} {float(v):.4f}" for k, v in gm.items()))
        out = os.path.join(cfg["output_dir"], "snorm_exps", f"{model.checkpoint_name}_{probe.name}".replace("$", ""))
        if not is_eval:
            print("saved", checkpoint.save_checkpoint(os.path.join(out, "ckpt.pth"), cfg, model, probe))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
