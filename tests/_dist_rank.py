"""One rank of tests/test_gpu_dist.py (started as a fresh interpreter per rank, torchrun-style environment).
Runs the PRODUCT step (DINO -> DepthHead -> DepthLoss -> FlatAdamW) for a few iterations on this rank's shard of the
synthetic stream and dumps the probe parameters + losses.  Several ranks share cuda:0 over gloo on a one-GPU box
(MVP_DIST_BACKEND=gloo MVP_FORCE_DEVICE=0); on a multi-GPU node the same code runs one rank per GPU over RCCL."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "midvision-probe_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

D, DEPTH, HEADS, B, H, W, STEPS = 128, 4, 2, 3, 64, 96, 3
PIPE_STEPS = 7  # mode "pipe": 21 images per rank = three span forwards of 7


def build(dev, probe_seed, overlap):
    from evals.models.dino import DINO
    from evals.models.probes import DepthHead
    from evals.utils.optim import cosine_decay_linear_warmup
    from mvp.optim import FlatAdamW
    from oracle import vit as ovit  # seeded tiny-ViT weights only (test infrastructure)

    vsd = ovit.make_vit_weights(embed_dim=D, depth=DEPTH, seed=41)
    model = DINO(return_multilayer=True, add_norm=True, weights=vsd).to(dev)
    torch.manual_seed(probe_seed)
    probe = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type="bindepth").to(dev)
    opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}], overlap_comm=overlap)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda e: cosine_decay_linear_warmup(e, 20, 2))
    return model, probe, opt, sched


def batch(rank, step, dev):
    from oracle import train as otrain

    images, tgt = otrain.synthetic_depth_batch(B, H, W, rank=rank, step=step)
    return images.to(dev), tgt.to(dev)


def main():
    out_dir, overlap = sys.argv[1], bool(int(sys.argv[2]))
    from evals.utils.losses import DepthLoss
    from mvp import dist as mdist
    from mvp.train import train_depth_step

    rank, local, world = mdist.env_setup("nccl")
    dev = torch.device("cuda", torch.cuda.current_device())
    # every rank seeds its probe DIFFERENTLY: equality afterwards proves the rank-0 broadcast of FlatAdamW.__init__
    model, probe, opt, sched = build(dev, probe_seed=100 + rank, overlap=overlap)
    loss_fn = DepthLoss()
    losses = []
    if len(sys.argv) > 3 and sys.argv[3] == "pipe":
        # the per-rank pipeline of a multi-GPU job with graphs switched on: span forwards (7 images over batches of 3) on a side stream,
        # (slot, carry) graphs captured up front and replayed, the all-reduce of step t pending while the forwards of later batches run
        from mvp.pipeline import FeaturePipeline, pipelined_features

        pipe = FeaturePipeline(model, 2, graphs=True, group=None, span=7)
        bs = [batch(rank, s, dev) for s in range(PIPE_STEPS)]
        for (img, tgt), f in pipelined_features(model, bs, pipe=pipe):
            losses.append(train_depth_step(model, probe, opt, sched, loss_fn, None, tgt, feats=f).item())
        assert pipe.span == 7 and pipe.graphs and sum(e["calls"] for e in pipe._graphs.values()) >= 2
    else:
        for s in range(STEPS):
            images, tgt = batch(rank, s, dev)
            losses.append(train_depth_step(model, probe, opt, sched, loss_fn, images, tgt).item())
    opt.finish_pending()
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), flat=opt.flat_param.cpu().numpy(), losses=np.array(losses),
             world=torch.distributed.get_world_size(), backend=np.array(torch.distributed.get_backend()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
