"""Child of tests/test_gpu_dist.py::test_rccl_world1_overlapped_allreduce_with_graph_replay: ONE rank, backend nccl (= RCCL), started
as a fresh interpreter before any GPU call.  Runs the product loop twice from identical states — (a) not distributed, (b) with
FlatAdamW(overlap_comm=True, force_comm=True): every step's flat gradient goes through a real RCCL all_reduce(async_op=True) on RCCL's
stream, the compute stream picks it up with work.wait(), AdamW is applied right before the next probe forward — with the grouped,
hipGraph-replayed frozen forwards of mvp/pipeline.py beside it.  Dumps both trajectories."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "midvision-probe_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(dist_on, steps=9):
    import _dist_rank as R
    from evals.utils.losses import DepthLoss
    from evals.utils.optim import cosine_decay_linear_warmup
    from mvp.optim import FlatAdamW
    from mvp.pipeline import FeaturePipeline, pipelined_features
    from mvp.train import train_depth_step

    dev = torch.device("cuda", torch.cuda.current_device())
    model, probe, _, _ = R.build(dev, probe_seed=100, overlap=False)
    opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}], overlap_comm=dist_on, force_comm=dist_on)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda e: cosine_decay_linear_warmup(e, 20, 2))
    bs = [R.batch(0, s, dev) for s in range(steps)]
    pipe = FeaturePipeline(model, 2, graphs=True, group=3)  # multi-rank jobs default to eager launches: graphs are forced ON here
    assert pipe.graphs
    losses = [train_depth_step(model, probe, opt, sched, DepthLoss(), None, tgt, feats=f).item() for (img, tgt), f in pipelined_features(model, bs, pipe=pipe)]
    opt.finish_pending()
    torch.cuda.synchronize()
    assert all(e["graph"] is not None for e in pipe._graphs.values()) and sum(e["calls"] for e in pipe._graphs.values()) >= 3
    return np.array(losses), opt.flat_param.cpu().numpy().copy(), opt.exp_avg_sq.cpu().numpy().copy()


def main():
    out = sys.argv[1]
    torch.cuda.set_device(0)
    a = run(False)
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    b = run(True)
    np.savez(out, la=a[0], pa=a[1], va=a[2], lb=b[0], pb=b[1], vb=b[2], backend=np.array(dist.get_backend()))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
