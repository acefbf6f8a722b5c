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


def self_launch(world: int, argv) -> int:
    """cfg.system.num_gpus > 1 without a torchrun environment: start one fresh rank per GPU BEFORE this process touches
    the GPU (the reference does mp.spawn(train_model, nprocs=world_size), train_depth.py:851-855)."""
    import socket
    import subprocess

    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main(argv):
    from evals.utils.losses import DepthLoss
    from evals.utils.optim import cosine_decay_linear_warmup
    from mvp import results

    cfg = config.compose("depth_training", argv)
    if int(cfg["system"]["num_gpus"]) > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(int(cfg["system"]["num_gpus"]), argv))
    if float(cfg["optimizer"].get("model_lr", 0.0)) != 0.0:
        raise NotImplementedError("optimizer.model_lr != 0 (backbone fine-tuning) is outside the frozen-backbone hot path")
    rank, local, world = mdist.env_setup("nccl")
    cfg["system"]["num_gpus"] = world
    torch.manual_seed(int(cfg["system"]["random_seed"]))  # probe initialisation; rank 0's copy is broadcast by FlatAdamW
    dev = torch.device("cuda", torch.cuda.current_device())
    from evals.datasets import build_loader

    ds = cfg["dataset"]
    loader = build_loader(dict(ds, batch_size=cfg["batch_size"]), "train", cfg["batch_size"], num_gpus=world, num_workers=cfg.get("num_workers", 2),
                          with_snorm=False)
    model = config.instantiate(cfg["backbone"]).to(dev)
    probe = config.instantiate(cfg["probe"], feat_dim=model.feat_dim, max_depth=ds["max_depth"]).to(dev)
    if "vit-mae" in model.checkpoint_name or "sam" in model.checkpoint_name:  # train_depth.py:613-617
        model.resize_pos_embed(image_size=tuple(loader.dataset[0]["image"].shape[-2:]))
    opt = FlatAdamW([{"params": probe.parameters(), "lr": cfg["optimizer"]["probe_lr"]}], overlap_comm=world > 1)
    n_ep = cfg["optimizer"]["n_epochs"]
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda e: cosine_decay_linear_warmup(
        e, n_ep * len(loader), cfg["optimizer"]["warmup_epochs"] * len(loader)))
    hist = []
    is_eval = bool(cfg.get("is_eval", False))
    loaded_ckpt = None
    if is_eval:
        # train_depth.py:526-535,570: an evaluation run scores the probe of cfg.ckpt_path (probe only: the reference's model line is
        # commented out); without a checkpoint it would score a randomly initialised probe under the real experiment's CSV columns
        loaded_ckpt = str(cfg.get("ckpt_path", "") or "").replace("\\$", "$")
        if not loaded_ckpt:
            raise SystemExit("is_eval=True needs ckpt_path=<.../ckpt.pth> (refusing to validate a randomly initialised probe)")
        checkpoint.load_checkpoint(loaded_ckpt, model, probe, load_model=False)
    else:
        hist = train(model, probe, loader, opt, sched, n_ep, detach_model=True, loss_fn=DepthLoss(), rank=rank, world_size=world)
    opt.finish_pending()
    if rank == 0:
        model.eval(); probe.eval()
        is_navi = ds["name"] in ("navi_reldepth", "navi")
        vcfg = dict(ds, num_batches=2, batch_size=cfg["batch_size"])
        timestamp, exp_name, exp_info = results.experiment_info(cfg, model, probe, ds["name"], ds["name"])
        out = os.path.join(cfg["output_dir"], "depth_exps", exp_name.replace("$", ""))
        ckpt = loaded_ckpt if is_eval else os.path.join(out, "ckpt.pth")  # the CSV's ckpt_path column names what was scored
        sa_loss, sa_g, sa_l = validate(model, probe, build_loader(vcfg, "valid", cfg["batch_size"], with_snorm=False), DepthLoss(), is_navi=is_navi)
        si_loss, si_g, si_l = validate(model, probe, build_loader(vcfg, "valid", cfg["batch_size"], with_snorm=False), DepthLoss(),
                                       scale_invariant=True, is_navi=is_navi)
        print(f"train loss/epoch {hist} | SA valid loss {sa_loss:.4f} d1 {sa_g['d1']:.4f} rmse {sa_g['rmse']:.4f} | SI loss {si_loss:.4f} rmse {si_g['rmse']:.4f}")
        titles, row = results.depth_result_row(timestamp, exp_info, sa_g, si_g, sa_l, si_l, ckpt, ds["name"])
        print("results ->", results.append_result_csv(results.result_csv_path(cfg["output_dir"], "depth", ds["name"], bool(cfg["backbone"].get("add_norm"))), titles, row))
        if not is_eval:
            print("saved", checkpoint.save_checkpoint(ckpt, cfg, model, probe))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
