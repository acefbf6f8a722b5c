#!/usr/bin/env python3
"""Host-side cost of the pipelined loop (diagnostic): cProfile over N steps of the headline loop (B = 16, 224^2, grouped forwards)."""
import cProfile, os, pstats, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from evals.models.dino import DINO
from evals.models.probes import DepthHead
from evals.utils.losses import DepthLoss
from evals.utils.optim import cosine_decay_linear_warmup
from mvp import backbone as bb, pipeline
from mvp.optim import FlatAdamW
from mvp.pipeline import FeaturePipeline, pipelined_features
from mvp.train import train_depth_step

dev = torch.device("cuda:0")
B = 16
model = DINO(return_multilayer=True, add_norm=True, weights=bb.random_vit_state_dict(seed=0)).to(dev)
torch.manual_seed(0)
probe = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type="bindepth", min_depth=0.001, max_depth=10).to(dev)
opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}])
sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda e: cosine_decay_linear_warmup(e, 10000, 150))
loss_fn = DepthLoss()
g = torch.Generator().manual_seed(0)
batches = [(torch.randn(B, 3, 224, 224, generator=g).to(dev), (torch.rand(B, 1, 224, 224, generator=g) * 9.9 + 0.05).to(dev)) for _ in range(4)]
pipeline.freeze_gc()
pipe = FeaturePipeline(model, 2, group=6, streams=1)


def loop(n):
    seq = [batches[i % 4] for i in range(n)]
    for (img, tgt), f in pipelined_features(model, seq, pipe=pipe):
        train_depth_step(model, probe, opt, sched, loss_fn, None, tgt, feats=f)


loop(12); torch.cuda.synchronize()
n = 240
t0 = time.perf_counter(); loop(n); th = time.perf_counter() - t0; torch.cuda.synchronize(); tw = time.perf_counter() - t0
print(f"no profiler: host enqueue {th / n * 1e3:.3f} ms/step, wall {tw / n * 1e3:.3f} ms/step")
# probe step only, features fixed: host cost without any device wait (queue far from full)
feats = None
for (img, tgt), f in pipelined_features(model, [batches[0]], pipe=FeaturePipeline(model, 1)):
    feats = f
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(100):
    train_depth_step(model, probe, opt, sched, loss_fn, None, batches[0][1], feats=feats)
th = time.perf_counter() - t0; torch.cuda.synchronize(); tw = time.perf_counter() - t0
print(f"probe step only: host {th / 100 * 1e3:.3f} ms/step, wall {tw / 100 * 1e3:.3f} ms/step")
pr = cProfile.Profile(); pr.enable(); loop(n); pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
