#!/usr/bin/env python3
"""Device timeline of the pipelined headline loop (diagnostic): eager launches with HIP events around EVERY entry-point
call, each recorded on the stream the kernel is launched on (mvp.lib.TRACE_CALLS).  Shows what rocprofv3 cannot — it serialises
dispatches across streams: the frozen forward of the next group of batches on its side stream and the probe steps of the current
group on the trainer's stream, on one clock.  Prints per stream the busy time, how much of the probe chain's kernel time runs under a
forward kernel, the effective duration of the forward's kernels beside / without probe kernels, and an excerpt of the timeline.

    python tools/micro/chain_timeline.py [G] [depth] [streams]      (default 6 2 1; B = 16, 224^2, bf16x3, linear bindepth probe)"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))
import torch
from evals.models.dino import DINO
from evals.models.probes import DepthHead
from evals.utils.losses import DepthLoss
from mvp import backbone as bb, lib
from mvp.optim import FlatAdamW
from mvp.pipeline import FeaturePipeline, freeze_gc, pipelined_features
from mvp.train import train_depth_step

G, depth, streams, SPAN = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 6), (2, 2), (3, 1), (4, 0)))  # SPAN: images per span forward (0: whole-batch groups of G)
dev = torch.device("cuda")
B = 16
model = DINO(return_multilayer=True, add_norm=True, weights=bb.random_vit_state_dict(seed=0)).to(dev)
torch.manual_seed(0)
probe = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type="bindepth", min_depth=0.001, max_depth=10).to(dev)
opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}])
loss_fn = DepthLoss()
g = torch.Generator().manual_seed(0)
batches = [(torch.randn(B, 3, 224, 224, generator=g).to(dev), (torch.rand(B, 1, 224, 224, generator=g) * 9.9 + 0.05).to(dev)) for _ in range(4)]
freeze_gc()
pipe = FeaturePipeline(model, depth, graphs=False, group=G, streams=streams, span=SPAN or None)


def run(n):
    for (img, tgt), f in pipelined_features(model, [batches[i % 4] for i in range(n)], pipe=pipe):
        train_depth_step(model, probe, opt, None, loss_fn, None, tgt, feats=f)


run(2 * G)
torch.cuda.synchronize()
n = 5 * G
t0 = time.perf_counter(); run(n); torch.cuda.synchronize(); untraced = (time.perf_counter() - t0) / n * 1e3
base = torch.cuda.Event(enable_timing=True)
base.record()
trace = []
lib.TRACE_CALLS = trace
t0 = time.perf_counter(); run(n); torch.cuda.synchronize(); traced = (time.perf_counter() - t0) / n * 1e3
lib.TRACE_CALLS = None
main_stream = torch.cuda.current_stream().cuda_stream
ev = sorted((base.elapsed_time(e0) * 1e3, base.elapsed_time(e1) * 1e3, name.replace("mvp_", ""), "probe" if st == main_stream else "fwd") for name, st, e0, e1 in trace)
t_lo = ev[len(ev) // 5][0]
t_hi = ev[4 * len(ev) // 5][0]
mid = [e for e in ev if t_lo <= e[0] <= t_hi]  # steady middle: groups 2-4 of 5
span = t_hi - t_lo
print(f"B={B} 224^2 bf16x3, " + (f"spans of {SPAN} images" if SPAN else f"{G} batches") + f" per frozen forward, depth {depth}, {streams} forward stream(s); {len(trace)} launches traced over {n} steps")
print(f"ms/step: {untraced:.3f} untraced, {traced:.3f} with an event pair around every launch")


def busy(evs):  # union length of intervals
    tot, end = 0.0, -1e30
    for a, b in sorted(evs):
        if b > end:
            tot += b - max(a, end)
            end = b
    return tot


fw = [(a, b) for a, b, n_, s in mid if s == "fwd"]
pr = [(a, b) for a, b, n_, s in mid if s == "probe"]
both = busy(fw) + busy(pr) - busy(fw + pr)
print(f"steady window {span / 1e3:.2f} ms: forward stream busy {busy(fw) / span:.3f}, trainer stream busy {busy(pr) / span:.3f}, "
      f"both at once {both / span:.3f}; {both / max(busy(pr), 1e-9):.2f} of the probe chain's kernel time runs under a forward kernel")
# effective duration of the forward's kernels with / without a probe kernel beside them
pr_sorted = sorted(pr)
stats = {}
for a, b, name, s in mid:
    if s != "fwd":
        continue
    ov = sum(max(0.0, min(b, pb) - max(a, pa)) for pa, pb in pr_sorted if pb > a and pa < b)
    k = stats.setdefault(name, [0, 0.0, 0, 0.0])
    if ov > 0.2 * (b - a):
        k[0] += 1; k[1] += b - a
    else:
        k[2] += 1; k[3] += b - a
print("forward kernels, mean us (beside a probe kernel for > 20 % of their time | alone):")
for name, (c1, t1, c0, t0_) in sorted(stats.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
    print(f"   {name:28s} beside: n={c1:4d} {t1 / max(c1, 1):8.1f} | alone: n={c0:4d} {t0_ / max(c0, 1):8.1f}")
ps = {}
for a, b, name, s in mid:
    if s == "probe":
        k = ps.setdefault(name, [0, 0.0])
        k[0] += 1; k[1] += b - a
print("probe-step kernels, mean us (all of them run while a forward is in flight): " + ", ".join(f"{n_} {t / c:.1f}" for n_, (c, t) in sorted(ps.items(), key=lambda kv: -kv[1][1])))
print("timeline excerpt (us since the window start; F = forward stream, P = trainer stream):")
t_ref = mid[len(mid) // 2][0]
shown = 0
for a, b, name, s in mid:
    if a < t_ref:
        continue
    print(f"   {'F' if s == 'fwd' else 'P'} {a - t_ref:9.1f} -> {b - t_ref:9.1f}  {name}")
    shown += 1
    if shown >= 70:
        break
