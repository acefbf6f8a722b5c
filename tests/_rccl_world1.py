"""Child of tests/test_gpu_dist.py::test_rccl_world1_overlapped_allreduce_with_graph_replay: ONE rank, backend nccl (= RCCL), started
as a fresh interpreter before any GPU call.  Runs the product loop twice from identical states — (a) not distributed, (b) with
FlatAdamW(overlap_comm=True, force_comm=True): every step's flat gradient goes through a real RCCL all_reduce(async_op=True) on RCCL's
stream, the compute stream picks it up with work.wait(), AdamW is applied right before the next probe forward — with the grouped,
hipGraph-replayed frozen forwards of mvp/pipeline.py beside it (argv[2] == "spans": span forwards with a carried batch and a ragged last
batch instead of whole groups).  Dumps both trajectories."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "midvision-probe_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(dist_on, steps=9, spans=False):
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
    if spans:
        # the shape of the pipeline an 8-GPU job runs (mvp/pipeline.py, "Spans"): forwards of 7 images over batches of 3 (every cut
        # position), the (slot, carry) graphs captured up front and replayed, a batch carried across forwards, then an epoch's ragged
        # last batch (2 images) — eager forwards of another shape — beside the all-reduce still pending from the previous step
        img, tgt = R.batch(0, steps, dev)
        bs.append((img[:2], tgt[:2]))
        pipe = FeaturePipeline(model, 2, graphs=True, group=None, span=7)
    else:
        pipe = FeaturePipeline(model, 2, graphs=True, group=3)  # (jobs with more than one rank default to eager launches: graphs are forced ON here)
    assert pipe.graphs
    losses = [train_depth_step(model, probe, opt, sched, DepthLoss(), None, tgt, feats=f).item() for (img, tgt), f in pipelined_features(model, bs, pipe=pipe)]
    opt.finish_pending()
    torch.cuda.synchronize()
    assert all(e["graph"] is not None for e in pipe._graphs.values()) and sum(e["calls"] for e in pipe._graphs.values()) >= 3
    if spans:
        assert pipe.span == 7 and len(pipe._graphs) == 6, (pipe.span, len(pipe._graphs))  # carries 0, 1, 2 on both slots
    return np.array(losses), opt.flat_param.cpu().numpy().copy(), opt.exp_avg_sq.cpu().numpy().copy()


def main():
    out = sys.argv[1]
    torch.cuda.set_device(0)
    spans = len(sys.argv) > 2 and sys.argv[2] == "spans"
    steps = 12 if spans else 9
    a = run(False, steps, spans)
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    b = run(True, steps, spans)
    np.savez(out, la=a[0], pa=a[1], va=a[2], lb=b[0], pb=b[1], vb=b[2], backend=np.array(dist.get_backend()))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
