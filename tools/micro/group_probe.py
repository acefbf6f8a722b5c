#!/usr/bin/env python3
"""Component times of the grouped headline step (diagnostic): the frozen forward of G stacked batches (ms per forward and per batch),
the probe step alone, and the whole pipelined loop at several lengths.  B = 16, 224^2, bf16x3, linear bindepth probe."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from evals.models.dino import DINO
from evals.models.probes import DepthHead
from evals.utils.losses import DepthLoss
from mvp import backbone as bb, pipeline
from mvp.optim import FlatAdamW
from mvp.pipeline import FeaturePipeline, pipelined_features
from mvp.train import train_depth_step, extract_features

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 16))
model = DINO(return_multilayer=True, add_norm=True, weights=bb.random_vit_state_dict(seed=0)).to(dev)
torch.manual_seed(0)
probe = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type="bindepth", min_depth=0.001, max_depth=10).to(dev)
opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}])
loss_fn = DepthLoss()
g = torch.Generator().manual_seed(0)
batches = [(torch.randn(B, 3, 224, 224, generator=g).to(dev), (torch.rand(B, 1, 224, 224, generator=g) * 9.9 + 0.05).to(dev)) for _ in range(4)]
pipeline.freeze_gc()


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# (a) forward of G stacked batches, one stream, eager (the pipeline context gives per-slot buffers and deferred BN updates)
for G in (1, 2, 3, 4, 5, 6, 8):
    imgs = torch.cat([batches[i % 4][0] for i in range(G)])

    def fwd():
        with pipeline._slot(0, 1, G):
            pipeline._take_deferred()
            extract_features(model, imgs)
            pipeline._take_deferred()
    ms = timed(fwd, 5)
    print(f"forward G={G}: {ms:7.3f} ms = {ms / G:6.3f} ms per batch = {G * B / ms * 1e3:7.0f} img/s forward-only", flush=True)

# (b) probe step alone (features of one batch given)
feats = extract_features(model, batches[0][0])
ms = timed(lambda: train_depth_step(model, probe, opt, None, loss_fn, None, batches[0][1], feats=feats), 20)
print(f"probe step alone: {ms:6.3f} ms", flush=True)

# (c) whole loop
for G, depth, streams, graphs in ((6, 2, 1, True), (6, 2, 1, False), (6, 2, 2, False), (5, 2, 1, False), (4, 2, 1, False), (4, 3, 2, False), (3, 3, 2, False)):
    pipe = FeaturePipeline(model, depth, graphs=graphs, group=G, streams=streams)
    for n in (20, 60):
        seq = [batches[i % 4] for i in range(n)]

        def loop():
            for (img, tgt), f in pipelined_features(model, seq, pipe=pipe):
                train_depth_step(model, probe, opt, None, loss_fn, None, tgt, feats=f)
        ms = timed(loop, 2)
        print(f"loop G={G} depth={depth} streams={streams} graphs={graphs} n={n}: {ms / n:6.3f} ms/step = {B * n / ms * 1e3:7.0f} img/s", flush=True)
