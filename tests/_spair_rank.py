"""One rank of tests/test_gpu_dist.py::test_spair_pair_sharding_two_ranks_equals_single_process (fresh interpreter per rank, torchrun-style
environment; two ranks share cuda:0 over gloo on the one-GPU pool).  Runs mvp.spair.evaluate_dataset on this rank's shard of a
synthetic SPair-shaped dataset — rank r takes pairs r, r + W, ...; one all_gather_object of the per-pair error vectors, re-sorted into
dataset order (evaluate_spair_correspondence.py:104-121) — and dumps the full-dataset recall and confusion matrix it returns."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "midvision-probe_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

D, DEPTH, PAIRS, SIZE, KPS = 128, 4, 19, 160, 9


def build(dev):
    from evals.models.ibot import iBOT
    from oracle import vit as ovit  # seeded tiny-ViT weights only (test infrastructure)

    return iBOT(add_norm=True, weights=ovit.make_vit_weights(embed_dim=D, depth=DEPTH, seed=56)).to(dev)


def dataset():
    from mvp import spair

    return spair.SyntheticSPair(num_pairs=PAIRS, image_size=SIZE, num_kps=KPS, seed=3)


def main():
    out_dir = sys.argv[1]
    from mvp import dist as mdist
    from mvp import spair

    rank, local, world = mdist.env_setup("nccl")
    dev = torch.device("cuda", torch.cuda.current_device())
    from mvp import pipeline

    model = build(dev)
    recall, conf = spair.evaluate_dataset(model, dataset(), 0.10, rank=rank, world=world)
    pipes = list(pipeline.cached_pipelines(model).values())
    np.savez(os.path.join(out_dir, f"spair{rank}.npz"), recall=np.float64(recall), conf=conf.numpy(), world=world,
             backend=np.array(torch.distributed.get_backend()), graphs=np.array([p.graphs for p in pipes]), depth=np.array([p.depth for p in pipes]),
             replays=np.array([sum(max(0, e["calls"] - 1) for e in p._graphs.values() if e.get("graph") is not None) for p in pipes]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
