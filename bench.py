#!/usr/bin/env python3
"""Headline benchmark: images/sec through ViT-B/16 multilayer (4 taps, train-mode tap BN)
feature extraction + linear depth-probe train step (DepthHead linear k=1 bindepth -> bilinear
upsample -> DepthLoss -> backward -> AdamW -> LambdaLR) on 224x224 synthetic batches.

    python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)

One "step" = the train_depth.py:99-143 loop body over one per-GPU batch already resident in
HBM.  Rank 0 prints ONE JSON line (contract in the task statement) carrying, besides the
throughput, a `roofline` object for the dominant kernel (the split-bf16 MFMA GEMM; algorithmic
2*M*N*K flops per launch / HIP-event duration on the launch stream) and a `cpu_baseline`
object (the CPU oracle — a port of the reference path — timed on the host cores on a bounded
sample of the same workload).
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time
import warnings

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "midvision-probe_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402


def _latest_profile(suffix: str):
    """Newest committed profiles/rNN_<suffix> (the PMC passes cannot run inside the timed process: rocprofv3 wraps it)."""
    import glob

    paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"r[0-9][0-9]_{suffix}")))
    return paths[-1] if paths else None


def pmc_traffic(kernel: str):
    """(memory-side bytes per launch of `kernel`, source file) from the newest committed PMC pass over this same default
    workload (tools/make_profiles.py), corrected as MI355X_MICROARCH.md prescribes for gfx950.  (None, None) if absent."""
    path = _latest_profile("pmc_traffic.json")
    try:
        with open(path) as f:
            per = json.load(f)["per_launch"]
    except (OSError, ValueError, KeyError, TypeError):
        return None, None
    hits = [v for k, v in per.items() if k.startswith(kernel.rstrip(">"))]
    if hits:  # (several instantiations may share the prefix — operand-layout variants of one kernel: the one with the most dispatches)
        return max(hits, key=lambda v: v.get("dispatches", 0))["hbm_bytes"], os.path.relpath(path, os.path.dirname(os.path.abspath(__file__)))
    return None, None


def pmc_mfma_busy(kernel: str):
    """SQ_VALU_MFMA_BUSY_CYCLES per launch of `kernel` from the newest committed profiles/rNN_pmc_summary.txt, or None."""
    import re

    path = _latest_profile("pmc_summary.txt")
    best = (-1, None)
    try:
        for line in open(path):
            if line.startswith(kernel.rstrip(">")):
                m = re.search(r"SQ_VALU_MFMA_BUSY_CYCLES=([0-9.e+]+)", line)
                n = re.search(r"n=\s*(\d+)", line)
                if m and n and int(n.group(1)) > best[0]:
                    best = (int(n.group(1)), float(m.group(1)))
    except (OSError, TypeError):
        return None
    return best[1]


def live_pmc(extra_args):
    """Memory-side traffic and MFMA-busy cycles of THIS invocation's workload, measured now: the parent — before it touches the GPU —
    runs `rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py <same workload flags> --pmc-child` twice (FETCH_SIZE alone:
    it takes 3 of the 4 TCC slots; then WRITE_SIZE + SQ_VALU_MFMA_BUSY_CYCLES), the way MI355X_MICROARCH.md prescribes (separate
    passes, --kernel-trace only).  Returns {kernel name: {"hbm_bytes", "mfma_busy"}} averaged per launch, or None (no rocprofv3 /
    a failed pass): bench.py then falls back to the newest committed profiles/rNN_pmc_* and says so."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile

    rp = shutil.which("rocprofv3")
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if rp is None or under_profiler:  # (a profiled parent already holds the GPU: no nested profiler runs)
        return None
    me = os.path.abspath(__file__)
    acc = {}
    try:
        for counters in (["FETCH_SIZE"], ["WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]):
            d = tempfile.mkdtemp(prefix="mvp_pmc_")
            cmd = [rp, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", d, "--", sys.executable, me, *extra_args, "--pmc-child"]
            # own session: on a timeout the whole group (rocprofv3 wrapper AND the profiled python) is ended, nothing is left on the GPU
            pr = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tempfile.gettempdir(),
                                  env=dict(os.environ, TMPDIR=tempfile.gettempdir()), start_new_session=True)
            try:
                pr.wait(timeout=150)
            except subprocess.TimeoutExpired:
                import signal

                os.killpg(pr.pid, signal.SIGKILL)
                pr.wait()
                return None
            files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            if pr.returncode != 0 or not files:
                return None
            for path in files:
                for row in csv.DictReader(open(path)):
                    name = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
                    name = re.sub(r"^void ", "", name)
                    name = re.sub(r"\(mvp_\w+( const\*)?(, [^)]*)?\)$", "", name)
                    e = acc.setdefault(name, {})
                    c = e.setdefault(row["Counter_Name"], [0.0, 0])
                    c[0] += float(row["Counter_Value"])
                    c[1] += 1
                    if row["Counter_Name"] == "GRBM_GUI_ACTIVE":  # the dispatch's duration under the same pass: the clock the chip held
                        dsum = e.setdefault("_dur_ns", [0.0, 0])
                        dsum[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                        dsum[1] += 1
            shutil.rmtree(d, ignore_errors=True)
    except Exception:
        return None
    out = {}
    for name, cs in acc.items():
        f = cs.get("FETCH_SIZE", [0.0, 1]); w = cs.get("WRITE_SIZE", [0.0, 1]); b = cs.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0, 1])
        ga = cs.get("GRBM_GUI_ACTIVE", [0.0, 0]); du = cs.get("_dur_ns", [0.0, 0])
        # gfx950: FETCH_SIZE is in KiB and reports half of wide coalesced reads (x2); WRITE_SIZE in KiB is exact;
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS): / 8 = busy cycles of the dispatch
        out[name] = {"hbm_bytes": round(2.0 * f[0] / max(f[1], 1) * 1024.0 + w[0] / max(w[1], 1) * 1024.0), "mfma_busy": b[0] / max(b[1], 1),
                     "gui_cycles": (ga[0] / ga[1] / 8.0) if ga[1] else None, "dur_us": (du[0] / du[1] / 1e3) if du[1] else None}
    return out


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step (reference default batch_size: 16)")
    ap.add_argument("--image-size", default="224", help="S or HxW (480x640 = BASELINE config #2)")
    ap.add_argument("--precision", default=os.environ.get("MVP_BENCH_PRECISION", "f16x2"), choices=["bf16x3", "f16x2", "bf16"],
                    help="arithmetic of the frozen ViT blocks' GEMMs.  f16x2 (the library's and this benchmark's default since round 4): two fp16 MFMA products per contraction over compensated fp16 pairs "
                         "(include/mvp_hip.h, MVP_PREC_F16X2: the weight's fp16 rounding error rides in the second product) — 1.5e-5 ... 2.3e-5 rel-L2 on every ViT-B/16 golden "
                         "of the reference, the same as bf16x3 (tests/test_gpu_kernels.py::test_vit_base_*: the contract is 1e-3); bf16x3: three bf16 products, the same error, "
                         "timed in the same run and reported as `precision_bf16x3`; bf16: one product, fails the contract (4-6e-3)")
    ap.add_argument("--no-alt-precision", action="store_true", help="skip the extra bf16x3 leg of an f16x2 run")
    ap.add_argument("--probe-precision", default="bf16x3", choices=["bf16x3", "bf16"],
                    help="arithmetic of the TRAINED probe's GEMMs / convolutions (fp32 master weights and AdamW either way).  bf16x3 = what the parity tests hold to the "
                         "reference; bf16 = one product, i.e. ordinary mixed-precision training of the probe on exact frozen features (secondary lines only)")
    ap.add_argument("--h2d", action="store_true", help="PCIe-inclusive variant: batches start in host memory and go through mvp.prefetch.DevicePrefetcher (not the headline value)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=6, help="timed CPU-oracle steps (~1.8 s each at B=16 on 16 cores: ~11 s bounded sample)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes (roofline.traffic / mfma_busy then come from the committed profiles/)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--sustained-steps", type=int, default=300, help="extra untimed-by-the-contract leg: the same step for this many more iterations (steady-state clocks); 0 = skip")
    ap.add_argument("--inflight", type=int, default=None,
                    help="frozen-backbone forwards kept in flight on side HIP streams (mvp/pipeline.py); 1 = one serial kernel chain; "
                         "default: mvp.pipeline.default_depth(probe) = what the trainers use (4 batches ahead on 3 streams under the linear probe, 1 under DPT)")
    ap.add_argument("--tiles", default="auto", choices=["auto", "alone", "shared"],
                    help="GEMM tile policy (mvp_gemm_args.tile_policy): auto = what the pipeline selects (shared-chip 128x128 tiles from 3 forwards in flight); "
                         "'shared' with --inflight 1 runs the pipelined run's kernels as one serial chain (profiling)")
    ap.add_argument("--group", type=int, default=None, help="batches stacked into one frozen forward (default: mvp.pipeline.default_group)")
    ap.add_argument("--span", type=int, default=None, help="images per frozen forward when forwards may end inside a batch (default: mvp.pipeline.default_span; 0 = whole batches only)")
    ap.add_argument("--no-serial-leg", action="store_true", help="skip the extra inflight=1 leg reported as pipeline.serial (profiling runs)")
    ap.add_argument("--prediction", default="bindepth", choices=["bindepth", "sigdepth"], help="bindepth = headline (256 bins); sigdepth = the reference's other depth predictor (secondary line)")
    ap.add_argument("--probe", default="linear", choices=["linear", "dpt"], help="linear = headline (k=1 bindepth); dpt = configs/probe/depth_dpt.yaml")
    return ap.parse_args()


def host_cores() -> int:
    """CPU cores this process may actually use: cgroup quota (the GPU box gives a 16-core share of
    a 256-core host) > affinity mask > os.cpu_count()."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return int(os.environ.get("MVP_CPU_THREADS", min(n, 16)))


def flops_per_image(N: int, D: int = 768, depth: int = 12, head_out: int = 256, taps: int = 4, probe: str = "linear", hidden: int = 512) -> float:
    """ViT fwd = depth*[N*14,155,776 + 4*N^2*768] + (N-1)*1,179,648; linear head
    (token resolution, conv1x1 and bilinear commute) fwd + dW = 2 * 2*(N-1)*taps*D*head_out.
    DPT probe (probes.py:215-399): 4 conv1x1 D->hidden at the token grid, 14 conv3x3 hidden->hidden at twice the grid,
    out_conv 3x3 hidden->hidden and 3x3 hidden->head_out at 8x the grid; backward = bwd-data + bwd-weight of each (no bwd-data
    into the detached features): ~142 GF forward, ~424 GF per image and step at 224^2.  These are the REFERENCE algorithm's flops (what
    `chip_level` divides by the step time: throughput in the reference's own arithmetic); since round 3 the library executes fewer — the
    3x3 convs that read a nearest-upsampled map run on the coarse grid (DESIGN §4 "Convolutions": out_conv[0] 947 -> 237 GF forward,
    2 x 947 -> 2 x 59 GF backward per 16 images) — so `chip_level` of a `--probe dpt` line is not an MFMA utilisation."""
    vit = depth * (N * (2 * (3 * D * D + D * D + 8 * D * D)) + 4 * N * N * D) + (N - 1) * 2 * D * D
    if probe == "dpt":
        hw = N - 1
        proj = taps * 2 * hw * D * hidden
        rcu = 14 * 2 * (4 * hw) * (9 * hidden) * hidden
        outc = 2 * (64 * hw) * (9 * hidden) * (hidden + head_out)
        return float(vit + 3 * (proj + rcu + outc) - proj)
    head = 2 * 2 * (N - 1) * taps * D * head_out
    return float(vit + head)


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a torchrun environment: start N fresh ranks (one per GPU) as children of this
    process BEFORE it has made any GPU call (importing torch does not initialise HIP), and return their exit code.
    Mirrors the reference's self-launch (mp.spawn over world_size, train_depth.py:851-855)."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _gemm_tiles_label(pipe, tiles_shared, B, H, W):
    """Which kernel family the backbone GEMMs of a forward run on (the mirror of pp_takes() in csrc/gemm.hip, mvp.ops.gemm_tile)."""
    from mvp import ops
    from mvp.pipeline import rows_per_image

    imgs = pipe.span or B * max(1, pipe.group or 1)
    M = imgs * rows_per_image(H, W, 16)
    fam = {ops.gemm_tile(M, n, k, tile_policy=1 if tiles_shared else 0) for n, k in ((2304, 768), (768, 768), (3072, 768), (768, 3072))}
    if all(t.startswith("pp ") for t in fam):
        return f"large-M 256x256 ping-pong kernel (gemm_pp.hip), M = {M} rows per forward"
    return ("shared-chip " if tiles_shared else "alone ") + " / ".join(sorted(fam)) + f" at M = {M}"


def main():
    args = parse_args()
    warnings.simplefilter("ignore")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    if args.pmc_child:  # the profiled child: a few steps of the workload, nothing else (steps are set once the group size is known)
        args.sustained_steps, args.no_cpu_baseline, args.no_roofline, args.no_live_pmc, args.no_serial_leg = 0, True, True, True, True
        os.environ["MVP_PIPELINE_GRAPHS"] = "0"  # eager launches: every kernel of the timed run's forwards appears in the trace
    pmc_live = None
    if not (args.no_live_pmc or args.no_roofline) and args.gpus == 1 and "WORLD_SIZE" not in os.environ:
        from mvp.pipeline import SHARED_TILES_FROM, default_depth

        class _ProbeName:  # default_depth only looks at the probe's name
            name = f"bindepth_{args.probe}_k"

        from mvp.pipeline import MAX_STREAMS

        # the counters are collected on the SAME pipeline configuration the timed run uses (grouped forwards, same kernel instantiations);
        # the profiler serialises the kernels, so every launch is alone on the chip
        wl = ["--batch", str(args.batch), "--image-size", args.image_size, "--precision", args.precision, "--probe", args.probe, "--probe-precision", args.probe_precision, "--tiles", args.tiles]
        if args.inflight is not None:
            wl += ["--inflight", str(args.inflight)]
        if args.group is not None:
            wl += ["--group", str(args.group)]
        if args.span is not None:
            wl += ["--span", str(args.span)]
        pmc_live = live_pmc(wl)  # before this process makes any GPU call
    from mvp import dist as mdist

    rank, local, world = mdist.env_setup("nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a job of a different size")
    backend = torch.distributed.get_backend() if world > 1 else "none"
    if world > 1:
        assert torch.distributed.get_world_size() == args.gpus
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    dev = torch.device("cuda", torch.cuda.current_device())
    if "x" in args.image_size:
        H, W = (int(v) for v in args.image_size.split("x"))
    else:
        H = W = int(args.image_size)
    B = args.batch

    from evals.models.dino import DINO
    from evals.models.probes import DepthHead
    from evals.utils.losses import DepthLoss
    from evals.utils.optim import cosine_decay_linear_warmup
    from mvp import backbone as bb
    from mvp import ops
    from mvp.optim import FlatAdamW
    from mvp.train import train_depth_step

    # random-init weights of the ViT-B/16 architecture (no network for checkpoints), same on every rank
    vsd = bb.random_vit_state_dict(seed=0)
    model = DINO(return_multilayer=True, add_norm=True, weights=vsd, precision=args.precision).to(dev)
    total_steps = args.warmup + args.steps + 8

    def make_probe():
        torch.manual_seed(0)
        if args.probe == "linear":
            pr = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type=args.prediction,
                           min_depth=0.001, max_depth=10, precision=args.probe_precision).to(dev)
        else:
            pr = DepthHead(feat_dim=model.feat_dim, head_type="dpt", kernel_size=3, prediction_type=args.prediction, hidden_dim=512,
                           min_depth=0.001, max_depth=10, precision=args.probe_precision).to(dev)
        # N > 1: rank 0's probe is broadcast at construction (DDP semantics); the flat-gradient all-reduce of step t runs
        # under the frozen forward of step t+1 and its AdamW update lands right before the probe forward
        op = FlatAdamW([{"params": pr.parameters(), "lr": 5e-4}], overlap_comm=world > 1)
        sc = torch.optim.lr_scheduler.LambdaLR(op, lr_lambda=lambda e: cosine_decay_linear_warmup(e, 10 * total_steps, 1.5 * total_steps))
        return pr, op, sc

    probe, opt, sched = make_probe()
    loss_fn = DepthLoss()

    # synthetic NYU-shaped batches, generated per (rank, step), resident in HBM before timing
    def make_batch(step):
        g = torch.Generator().manual_seed(1000 * rank + step)
        images = torch.randn(B, 3, H, W, generator=g)
        depth = torch.rand(B, 1, H, W, generator=g) * 9.9 + 0.05
        depth[torch.rand(B, 1, H, W, generator=g) < 0.1] = 0.0
        return images.to(dev), depth.to(dev)

    n_distinct = 4
    batches = [make_batch(s) for s in range(n_distinct)]

    def step(i):
        images, target = batches[i % n_distinct]
        return train_depth_step(model, probe, opt, sched, loss_fn, images, target)

    from mvp.pipeline import FeaturePipeline

    from mvp.pipeline import SHARED_TILES_FROM, default_depth, shared_tiles

    # forwards in flight: --inflight (None = the trainers' default: mvp.pipeline.default_depth / default_group), --group batches per forward
    d0 = default_depth(probe)
    d_final = os.environ.get("MVP_INFLIGHT") is not None or os.environ.get("MVP_FORCE_DEVICE") is not None
    pipe = FeaturePipeline(model, args.inflight if args.inflight is not None else (d0 if d_final else None),
                           group=args.group, span=args.span, ungrouped_depth=d0 if args.inflight is None else None)
    pipe.resolve_group(batches[0][0])
    if args.pmc_child:
        args.warmup, args.steps = pipe.group, ((2 * pipe.span) // B if pipe.span else 2 * pipe.group)  # about two full forwards of the timed run's shape
    if args.tiles == "alone" and pipe.chains >= SHARED_TILES_FROM:
        raise SystemExit("--tiles alone contradicts --inflight >= 3 (the pipeline selects the shared-chip tiles)")
    # the tile policy of the timed run's backbone GEMMs; a serial chain (--inflight 1) can be forced to it for profiling
    tiles_shared = pipe.chains >= SHARED_TILES_FROM or args.tiles == "shared"
    force_shared = args.tiles == "shared"

    from mvp.pipeline import pipelined_features

    def run_steps(i0, n, out=None, pipe=pipe, objs=None, model=model):
        """Steps i0 .. i0+n-1, every one the full train_depth.py:99-143 body.  The frozen forwards of upcoming batches (stacked ``group``
        at a time) are in flight on side streams while the probe steps run in order on this stream; the pipeline starts empty and ends
        empty, so all the work of these n steps (the forwards of n batches, n probe forward/backward/AdamW) lies between the caller's two
        barriers."""
        seq = [batches[i % n_distinct] for i in range(i0, i0 + n)]
        pr_, op_, sc_ = objs if objs is not None else (probe, opt, sched)
        ctx = shared_tiles(True) if (pipe.depth == 1 and force_shared) else contextlib.nullcontext()  # (a pipelined forward selects its policy itself)
        with ctx:
            for (images, target), feats in pipelined_features(model, seq, pipe=pipe):
                loss = train_depth_step(model, pr_, op_, sc_, loss_fn, None, target, feats=feats)
                if out is not None:
                    out.append(loss)

    def barrier():
        opt.finish_pending()  # the last step's update belongs to the timed region
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ---------------- in-run check (VERDICT r2 #2c): from identical probe / optimizer states, the pipelined loop (grouped forwards, graph
    # replay) and the one-batch-at-a-time serial loop produce the same per-step losses, bit for bit.  (Also sets the pipeline's graphs up.)
    pipeline_check = None
    if pipe.depth > 1 and not args.pmc_child and world == 1:
        nchk = pipe.group + min(2, pipe.group)  # one full group + a ragged one
        la, lb = [], []
        run_steps(0, nchk, la, pipe, make_probe())
        torch.cuda.synchronize()
        run_steps(0, nchk, lb, FeaturePipeline(model, 1), make_probe())
        torch.cuda.synchronize()
        same = bool(torch.equal(torch.stack(la), torch.stack(lb)))
        pipeline_check = {"steps": nchk, "losses_bit_identical_to_serial_loop": same}
        assert same, "the pipelined loop's losses differ from the serial loop's"

    from mvp.pipeline import freeze_gc

    freeze_gc()  # as the trainers do (mvp/train.py): no full-heap collector pause (75-110 ms) in the middle of a leg
    losses = []  # device scalars, reduced after the timed region (no accumulate kernel inside it, no host sync per step)
    run_steps(0, args.warmup, losses)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps, losses)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    last_loss = float(torch.stack(losses).mean().item())
    loss_acc = torch.zeros((), device=dev)
    images_per_s = world * B * args.steps / dt

    # ---------------- serial leg (reported next to `value`): the same steps as ONE kernel chain on one stream (inflight = 1)
    pipeline_info = {"inflight": pipe.depth, "group": pipe.group, "span_images": (pipe.span or None), "streams": pipe.chains, "hipgraph_forward": pipe.graphs, "gemm_tiles": _gemm_tiles_label(pipe, tiles_shared, B, H, W),
                     "what": "frozen forwards of upcoming batches (stacked `group` at a time into one chain of launches: same bits per batch) run on side "
                             "HIP streams under the probe steps of the current batches; every step still runs its own full forward + probe "
                             "forward/backward/AdamW inside the timed region, and the pipeline is empty at both of its barriers"}
    pipeline_info["check"] = pipeline_check
    pipeline_info["probe_step"] = ("tape-free: the autograd path's launches issued without a tape (mvp/fused_step.py; bit-identical trajectories)"
                                   if getattr(opt, "_mvp_fused_plan", None) not in (None, False) else "autograd tape")
    if pipe.depth > 1 and not args.no_serial_leg:
        serial_pipe = FeaturePipeline(model, 1)
        barrier()
        ts = time.perf_counter()
        run_steps(args.warmup + args.steps, args.steps, None, serial_pipe)
        barrier()
        sdt1 = time.perf_counter() - ts
        if world > 1:
            tm = torch.tensor([sdt1], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tm, op=torch.distributed.ReduceOp.MAX)
            sdt1 = float(tm.item())
        pipeline_info["serial"] = {"steps": args.steps, "value": round(world * B * args.steps / sdt1, 2), "unit": "images/s",
                                   "ms_per_step": round(sdt1 / args.steps * 1e3, 4)}

    # ---------------- sustained leg (reported next to `value`, never instead of it): a short run sits in boost clocks
    sustained = None
    if args.sustained_steps > 0:
        barrier()
        w0 = pipe.throttle_wait_s
        t1 = time.perf_counter()
        run_steps(args.warmup + args.steps, args.sustained_steps)
        host_dt = time.perf_counter() - t1  # host enqueue time of the leg (the device runs behind it) ...
        host_wait = pipe.throttle_wait_s - w0  # ... of which the host spent this much waiting in the pipeline's run-ahead throttle
        barrier()
        sdt = time.perf_counter() - t1
        if world > 1:
            tm = torch.tensor([sdt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tm, op=torch.distributed.ReduceOp.MAX)
            sdt = float(tm.item())
        sustained = {"steps": args.sustained_steps, "value": round(world * B * args.sustained_steps / sdt, 2), "unit": "images/s",
                     "ms_per_step": round(sdt / args.sustained_steps * 1e3, 4), "host_enqueue_ms_per_step": round(host_dt / args.sustained_steps * 1e3, 4),
                     # the enqueue time split: waiting for the device in FeaturePipeline's run_ahead throttle / actual host work (Python,
                     # ctypes, autograd, launch calls): the second is what a busy host or eight ranks per node must afford per step
                     "host_throttle_wait_ms_per_step": round(host_wait / args.sustained_steps * 1e3, 4),
                     "host_work_ms_per_step": round((host_dt - host_wait) / args.sustained_steps * 1e3, 4)}

    # ---------------- the same timed region in the three-product arithmetic (MVP_PRECISION=bf16x3): reported next to `value`, never instead of it
    alt_precision = None
    if args.precision == "f16x2" and not args.no_alt_precision and not args.pmc_child and world == 1:
        model_b = DINO(return_multilayer=True, add_norm=True, weights=vsd, precision="bf16x3").to(dev)
        pipe_b = FeaturePipeline(model_b, args.inflight if args.inflight is not None else (d0 if d_final else None),
                                 group=args.group, span=args.span, ungrouped_depth=d0 if args.inflight is None else None)
        objs_b = make_probe()
        run_steps(0, args.warmup, None, pipe_b, objs_b, model_b)
        barrier()
        tb = time.perf_counter()
        run_steps(args.warmup, args.steps, None, pipe_b, objs_b, model_b)
        objs_b[1].finish_pending()
        barrier()
        bdt = time.perf_counter() - tb
        alt_precision = {"precision": "bf16x3", "steps": args.steps, "value": round(B * args.steps / bdt, 2), "unit": "images/s", "ms_per_step": round(bdt / args.steps * 1e3, 4),
                         "note": "MVP_PRECISION=bf16x3 (three bf16 MFMA products per contraction, fp32's exponent range; the same 1.5e-5 ... 2.3e-5 on the reference's ViT-B/16 goldens as f16x2), "
                                 "same pipeline, same timed-region rules, same process"}
        del model_b, pipe_b, objs_b
        torch.cuda.empty_cache()

    # ---------------- optional PCIe-inclusive leg (never `value`): the same steps fed from HOST memory
    h2d = None
    if args.h2d:
        from mvp.prefetch import DevicePrefetcher

        def host_batch(step_idx):
            g = torch.Generator().manual_seed(1000 * rank + step_idx)
            images = torch.randn(B, 3, H, W, generator=g)
            depth = torch.rand(B, 1, H, W, generator=g) * 9.9 + 0.05
            depth[torch.rand(B, 1, H, W, generator=g) < 0.1] = 0.0
            return {"image": images, "depth": depth}

        hb = [host_batch(s) for s in range(n_distinct)]
        res = {}
        for label, pinned in (("pageable", False), ("pinned", True)):
            src = [{k: (v.pin_memory() if pinned else v) for k, v in b.items()} for b in hb]
            from mvp.pipeline import pipelined_features

            # as mvp.train.train() does: prefetcher (side-stream H2D into recycled device buffers) -> forwards in flight -> probe steps.
            # Two separate passes, each from an EMPTY pipeline to an empty one (the rule of the headline's timed region): a warm-up pass,
            # then the timed pass — were the clock started in the middle of one pass, the span forwards already submitted for the next
            # ~7 batches would fall outside it (round 3's figures of this leg were taken that way, with single-batch forwards: a smaller error).
            def h2d_pass(n):
                seq = [src[i % n_distinct] for i in range(n)]
                acc = torch.zeros((), device=dev)
                for b, feats in pipelined_features(model, DevicePrefetcher(seq, dev, depth=2), depth=pipe.depth):
                    acc += train_depth_step(model, probe, opt, sched, loss_fn, None, b["depth"], feats=feats)
                return acc

            h2d_pass(max(args.warmup, 8))
            barrier()
            t1 = time.perf_counter()
            loss_acc += h2d_pass(args.steps)
            barrier()
            res[label] = round(world * B * args.steps / (time.perf_counter() - t1), 1)
        sync_t = time.perf_counter()
        for i in range(args.steps):  # the reference's way: blocking pageable .to(device) inside the step (train_depth.py:102-104)
            b = hb[i % n_distinct]
            loss_acc += train_depth_step(model, probe, opt, sched, loss_fn, b["image"].to(dev), b["depth"].to(dev))
        barrier()
        res["blocking_to_device"] = round(world * B * args.steps / (time.perf_counter() - sync_t), 1)
        h2d = {"unit": "images/s", "note": "inputs start in host memory; DevicePrefetcher(depth=2) feeding the pipeline (as mvp.train.train does) vs a blocking .to(device) per step on the serial loop", **res}

    gh, gw = -(-H // 16), -(-W // 16)
    N = 1 + gh * gw
    f_img = flops_per_image(N, probe=args.probe)

    # ---------------- roofline leg: per-launch HIP-event timing of every traced kernel, in the TIMED regime (the same pipeline shape,
    # forwards on their side stream, probe steps beside them; eager launches so that each launch can be bracketed by events recorded on
    # the stream it is launched on), then the forward's kernels once more with nothing beside them ("kernel_alone")
    roofline = None
    if not args.no_roofline:
        # every rank runs these extra steps (the optimiser step holds the gradient all-reduce); rank 0 reports
        i_rl = args.warmup + args.steps
        if pipe.depth > 1:
            eager = FeaturePipeline(model, pipe.depth, graphs=False, group=pipe.group, streams=pipe.chains, span=pipe.span)
            nrl = (2 * pipe.span) // B if pipe.span else 2 * pipe.group
        else:
            eager, nrl = FeaturePipeline(model, 1), 3
        run_steps(i_rl, nrl, None, eager)  # (allocates the eager pipeline's slot buffers)
        barrier()
        trace = []
        ops.set_trace(trace)
        with shared_tiles(tiles_shared) if pipe.depth == 1 else contextlib.nullcontext():
            run_steps(i_rl, nrl, None, eager)
        barrier()
        alone = []
        ops.set_trace(alone)
        nalone = 0
        if pipe.depth > 1:  # forwards only, one after the other: every kernel alone on the chip
            for _ in range(2):
                if pipe.span:  # one full span from a batch boundary (the last batch cut)
                    T = pipe.span
                    pieces = [batches[i % n_distinct][0] for i in range(-(-T // B))]
                    pieces[-1] = pieces[-1][:T - (len(pieces) - 1) * B]
                    eager.submit_span(pieces, B, 0)
                    nb = T // B
                else:
                    eager.submit_group([batches[i % n_distinct][0] for i in range(pipe.group)])
                    nb = pipe.group
                for _ in range(nb):
                    eager.next()
                nalone += nb
            eager.drain()
        barrier()
        ops.set_trace(None)
    if not args.no_roofline and rank == 0:
        def aggregate(tr):
            groups, hbm = {}, {}
            for kind, tile, prec, flops, e0, e1 in tr:
                gsum = (hbm if kind == "hbm" else groups).setdefault((kind, tile), [0.0, 0.0, 0])
                gsum[0] += flops
                gsum[1] += e0.elapsed_time(e1) * 1e-3
                gsum[2] += 1
            return groups, hbm

        groups, hbm = aggregate(trace)
        groups_alone, _ = aggregate(alone)
        # dominant = the kernel instantiation with the largest total time
        dom = max(groups.items(), key=lambda kv: kv[1][1])
        (kind, tile), (fl, sec, cnt) = dom
        achieved = fl / sec / 1e12
        name = ("gemm_pp_kernel<" if tile.startswith("pp ") else "gemm_kernel<") if kind == "gemm" else "attention_kernel"
        full_name = (("gemm_pp_kernel<" + tile[3:] + ">") if tile.startswith("pp ") else f"gemm_kernel<{tile}>") if kind == "gemm" else "attention_kernel"
        default_wl = (B, H, W, args.precision, args.probe) == (16, 224, 224, "bf16x3", "linear")
        traffic, traffic_src, busy, gui, pdur = None, None, None, None, None
        if pmc_live:
            hit = [v for k, v in pmc_live.items() if k.startswith(name if tile.startswith("pp ") else full_name.rstrip(">"))]
            if hit:
                best = max(hit, key=lambda v: v["mfma_busy"] or 0.0)  # (the instantiation that did the work: layout variants share the prefix)
                traffic, busy, traffic_src = best["hbm_bytes"], best["mfma_busy"] or None, "live: rocprofv3 --pmc child passes of this invocation"
                gui, pdur = best.get("gui_cycles"), best.get("dur_us")
        if traffic is None and default_wl:
            traffic, traffic_src = pmc_traffic(name if tile.startswith("pp ") else full_name)
            busy = pmc_mfma_busy(name if tile.startswith("pp ") else full_name)
            traffic_src = f"committed pass {traffic_src}"
        ka = groups_alone.get((kind, tile))
        sec_alone = (ka[1] / ka[2]) if ka else None
        clk = 1.97e9 if tile.startswith("pp ") else 2.1e9  # clock the chip holds under this kernel when no GRBM_GUI_ACTIVE pass is at hand
        roofline = {
            "bound": "mfma", "achieved": round(achieved, 2), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(achieved / 2500.0, 4),
            "traffic": traffic, "traffic_unit": f"bytes/launch (memory-side, rocprofv3 PMC: FETCH_SIZE x2 + WRITE_SIZE; {traffic_src})",
            "mfma_busy": None if busy is None else (
                {"SQ_VALU_MFMA_BUSY_CYCLES_per_launch": busy, "frac_of_simd_cycles": round(busy / (1024 * gui), 3), "clock_ghz_in_that_pass": round(gui / (pdur * 1e3), 3),
                 "note": "same PMC pass: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); the clock = those cycles / the dispatch's duration in that pass"}
                if gui and pdur else
                {"SQ_VALU_MFMA_BUSY_CYCLES_per_launch": busy, "frac_of_simd_cycles": round(busy / (1024 * (sec_alone or sec / cnt) * clk), 3),
                 "note": f"PMC pass (see traffic_unit); 1024 SIMDs x launch duration alone on the chip x {clk / 1e9:.2f} GHz (GRBM_GUI_ACTIVE / 8 / duration, profiles/r03_pmc_summary.txt)"}),
            "hbm_kernels": {k[1]: {"launches_per_step": round(v[2] / nrl, 2), "avg_us": round(v[1] / v[2] * 1e6, 2), "alg_mbytes_per_launch": round(v[0] / v[2] / 1e6, 2),
                                   "achieved_gbps": round(v[0] / v[1] / 1e9, 1), "frac_of_8TBps": round(v[0] / v[1] / 8e12, 3)} for k, v in hbm.items()},
            "kernel": full_name, "launches_per_step": round(cnt / nrl, 2), "avg_launch_us": round(sec / cnt * 1e6, 2),
            "alg_gflop_per_launch": round(fl / cnt / 1e9, 3),
            "regime": "timed: HIP events around every launch (on the stream it is launched on) while the pipeline runs as in the timed region — "
                      f"{pipe.group} batches per frozen forward on {pipe.chains} side stream(s), the probe steps of the previous batches beside it; eager launches instead of graph replay",
            "note": ("algorithmic 2*M*N*K flops (bf16x3 issues 3 MFMA passes per algorithmic flop: x3 = share of the bf16 MFMA pipe)" if args.precision == "bf16x3" else
                     "algorithmic 2*M*N*K flops (f16x2 issues 2 f16 MFMA passes — the bf16 pipe's rate — per algorithmic flop: x2 = share of the MFMA pipe)" if args.precision == "f16x2" else
                     "algorithmic 2*M*N*K flops, one bf16 MFMA pass each"),
            "kernel_alone": None if not ka else {"avg_launch_us": round(ka[1] / ka[2] * 1e6, 2), "alg_tflops": round(ka[0] / ka[1] / 1e12, 2), "frac": round(ka[0] / ka[1] / 1e12 / 2500.0, 4),
                                                 "note": "the same launches with nothing beside them (forwards only, one stream): what rocprofv3 --kernel-trace reports too"},
            "chip_level": {"alg_tflops": round(images_per_s / world * f_img / 1e12, 2), "frac": round(images_per_s / world * f_img / 1e12 / 2500.0, 4),
                           "note": "whole-step algorithmic flops / measured step time of the timed run"},
            "all_kernels": {f"{k[0]}:{k[1]}": {"launches_per_step": round(v[2] / nrl, 2), "avg_us": round(v[1] / v[2] * 1e6, 2),
                                                "alg_tflops": round(v[0] / v[1] / 1e12, 2)} for k, v in groups.items()},
            "whole_step_alg_tflops": round(images_per_s / world * f_img / 1e12, 2),
        }

    # ---------------- CPU baseline leg: the oracle (port of the reference CPU path) on the host cores
    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1 and args.probe == "linear":
        from oracle import probes as oprobes
        from oracle import train as otrain

        cores = host_cores()
        torch.set_num_threads(cores)
        psd = {"head.conv.weight": probe.head.conv.weight.detach().cpu().clone(), "head.conv.bias": probe.head.conv.bias.detach().cpu().clone()}
        tr = otrain.DepthProbeTrainer(vsd, psd, max_step=1000, warmup_step=150)
        Bc = B
        ci, ct = otrain.synthetic_depth_batch(Bc, H, W, rank=0, step=0)
        tr.step(ci, ct.clone())  # warm-up (thread pools, allocator)
        t0 = time.perf_counter()
        for s in range(args.cpu_steps):
            ci, ct = otrain.synthetic_depth_batch(Bc, H, W, rank=0, step=1 + s)
            tr.step(ci, ct)
        cdt = time.perf_counter() - t0
        cpu = {"value": round(Bc * args.cpu_steps / cdt, 3), "unit": "images/s", "cores": cores, "kind": "port",
               "sample": f"{args.cpu_steps} steps of B={Bc} {H}x{W} (same step: extract 4 taps + linear bindepth probe + DepthLoss + backward + AdamW), 1 warm-up step, torch {torch.__version__} CPU fp32"}
        # BASELINE.md §3's other points, on bounded samples: B = 8 (config #1's batch) on all cores — extract only and the whole step —
        # and one thread (a single scalar-port core, the "cores: 1" reading of the same path)
        variants = {}
        ci8, ct8 = otrain.synthetic_depth_batch(8, H, W, rank=0, step=100)
        tr.step(ci8, ct8.clone())
        t0 = time.perf_counter()
        for s_ in range(3):
            tr.features(ci8)
        variants["extract_only_B8"] = {"value": round(8 * 3 / (time.perf_counter() - t0), 3), "cores": cores, "sample": "3 frozen forwards (4 taps) of B=8"}
        t0 = time.perf_counter()
        for s_ in range(3):
            tr.step(ci8, ct8.clone())
        variants["step_B8"] = {"value": round(8 * 3 / (time.perf_counter() - t0), 3), "cores": cores, "sample": "3 steps of B=8, 1 warm-up"}
        torch.set_num_threads(1)
        ci1, ct1 = otrain.synthetic_depth_batch(2, H, W, rank=0, step=200)
        tr.step(ci1, ct1.clone())  # warm-up at one thread
        t0 = time.perf_counter()
        tr.step(ci1, ct1.clone())
        variants["step_1thread"] = {"value": round(2 / (time.perf_counter() - t0), 3), "cores": 1, "sample": "1 step of B=2 on one thread, 1 warm-up"}
        torch.set_num_threads(cores)
        cpu["variants"] = variants

    # ---------------- BASELINE config #1 (CPU-only by definition): DINO ResNet-50 random-init, single last-stage tap, 8 x 224 x 224
    # -> internally resized to 480^2 -> [8, 2048, 15, 15]; the oracle port of that path on the host cores, bounded sample
    cpu1 = None
    if not args.no_cpu_baseline and rank == 0 and world == 1 and args.probe == "linear" and (H, W) == (224, 224):
        from oracle import resnet as ores

        cores = host_cores()
        torch.set_num_threads(cores)
        rsd = ores.make_resnet50_weights(seed=0)
        g1 = torch.Generator().manual_seed(0)
        imgs = torch.randn(8, 3, 224, 224, generator=g1)
        with torch.no_grad():
            ores.resnet_dense_features(rsd, imgs, [4], fixed_size=480)  # warm-up
            t0 = time.perf_counter()
            n1 = 3
            for _ in range(n1):
                f4 = ores.resnet_dense_features(rsd, imgs, [4], fixed_size=480)
            c1dt = time.perf_counter() - t0
        cpu1 = {"value": round(8 * n1 / c1dt, 3), "unit": "images/s", "cores": cores, "kind": "port", "output_shape": list(f4.shape),
                "sample": f"{n1} forward passes of DINO ResNet-50 (random-init, single tap, add_norm) on 8x224x224 -> 480^2, 1 warm-up, torch {torch.__version__} CPU fp32"}

    if rank == 0:
        out = {
            "metric": "images/sec feature-extract+probe-step, ViT-B/16 224^2" if (H, W) == (224, 224) else f"images/sec feature-extract+probe-step, ViT-B/16 {H}x{W}",
            "value": round(images_per_s, 2), "unit": "images/s",
            # a gloo run with MVP_FORCE_DEVICE is a control-flow rehearsal of several ranks on ONE card: it is not an N-GPU number
            "n_gpus": 1 if (backend == "gloo" and os.environ.get("MVP_FORCE_DEVICE") is not None) else world,
            "ranks": world, "rccl_ranks": world if backend == "nccl" else 0,
            "dist_backend": backend, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16x3": "bf16x3 (split-bf16 MFMA, three products per contraction, fp32 accumulate; fp32 residual/LN/softmax/loss)",
                      "f16x2": "f16x2 (compensated fp16 pairs of activations and frozen weights, two f16 MFMA products per contraction, fp32 accumulate, feature error as bf16x3; attention: Q.K^T "
                               "the same two products over compensated pairs, P.V fp16 probabilities x (fp16 + bf16) V; probe, fp32 residual/LN/softmax/loss as in bf16x3)",
                      "bf16": "bf16 (MFMA, fp32 accumulate)"}[args.precision],
            "data": "synthetic (randn images, U(0.05,9.95) depth with 10% zeros), random-init ViT-B/16",
            "config": {"workload": f"dino_vitb16 return_multilayer(4 taps, add_norm train-mode BN) {H}x{W} + DepthHead(" + ("linear,k=1" if args.probe == "linear" else "dpt,k=3,hidden512") + f",{args.prediction}) + bilinear upsample + DepthLoss + backward + AdamW + LambdaLR",
                       "per_gpu_batch": B, "global_batch": B * world, "tokens_per_image": N, "parallelism": f"dp{world}",
                       "precision": args.precision, "probe_precision": args.probe_precision, "alg_gflop_per_image": round(f_img / 1e9, 2)},
            "mean_loss": round(last_loss, 5),
            "pipeline": pipeline_info,
            "sustained": sustained,
            **({"precision_bf16x3": alt_precision} if alt_precision else {}),
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_config1": cpu1, **({"h2d_inclusive": h2d} if h2d else {}),
        }
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
