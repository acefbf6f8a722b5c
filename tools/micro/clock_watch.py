#!/usr/bin/env python3
"""Diagnostic: sample the card's shader clock / power from sysfs (hwmon freq1_input, power1_average|power1_input) every few ms in a child
process while alternating pipelined and serial legs of the headline step.  Answers: are the sporadic slow legs clock / power events?"""
import glob, multiprocessing as mp, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "midvision-probe_amd"))


def find_files():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input", "temp2_input"):
            p = os.path.join(hw, name)
            if os.path.exists(p):
                out.setdefault(hw, {})[name] = p
    return out


def sampler(stop, path_sets, out_path, period):
    with open(out_path, "w") as f:
        while not stop.is_set():
            t = time.perf_counter()
            vals = []
            for hw, files in path_sets.items():
                for name, p in files.items():
                    try:
                        vals.append(f"{name}={open(p).read().strip()}")
                    except OSError as e:
                        vals.append(f"{name}=ERR")
            f.write(f"{t:.4f} " + " ".join(vals) + "\n")
            f.flush()
            time.sleep(period)


if __name__ == "__main__":
    files = find_files()
    print("hwmon files:", {k: sorted(v) for k, v in files.items()}, flush=True)
    # keep only hwmon dirs of the first card that has a freq file
    stop = mp.Event()
    log = os.path.join(REPO, "gpurun_out", "clock_watch_samples.txt")
    os.makedirs(os.path.dirname(log), exist_ok=True)
    proc = mp.get_context("spawn").Process(target=sampler, args=(stop, files, log, 0.004))
    proc.start()

    import torch
    from evals.models.dino import DINO
    from evals.models.probes import DepthHead
    from evals.utils.losses import DepthLoss
    from mvp import backbone as bb
    from mvp.optim import FlatAdamW
    from mvp.pipeline import FeaturePipeline
    from mvp.train import train_depth_step

    dev = torch.device("cuda")
    B = 16
    model = DINO(return_multilayer=True, add_norm=True, weights=bb.random_vit_state_dict(seed=0)).to(dev)
    probe = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type="bindepth", min_depth=0.001, max_depth=10).to(dev)
    opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}])
    loss_fn = DepthLoss()
    batches = [(torch.randn(B, 3, 224, 224, device=dev), torch.rand(B, 1, 224, 224, device=dev) * 9 + 0.05) for _ in range(4)]

    def run(pipe, n):
        nxt = 0
        for i in range(n):
            while len(pipe) < pipe.depth and nxt < n:
                pipe.submit(batches[nxt % 4][0])
                nxt += 1
            train_depth_step(model, probe, opt, None, loss_fn, None, batches[i % 4][1], feats=pipe.next())

    pipes = {1: FeaturePipeline(model, 1, run_ahead=0), 2: FeaturePipeline(model, 2, run_ahead=0)}
    run(pipes[2], 10)
    torch.cuda.synchronize()
    legs = []
    plan = [(2, 30), (1, 30), (2, 100), (1, 100), (2, 300), (1, 300), (2, 30), (1, 100), (2, 100), (1, 30)] * 2
    for depth, n in plan:
        t0 = time.perf_counter()
        run(pipes[depth], n)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        legs.append((depth, n, t0, t1))
    stop.set()
    proc.join(5)
    samples = []
    for line in open(log):
        parts = line.split()
        d = {kv.split("=")[0]: kv.split("=")[1] for kv in parts[1:]}
        samples.append((float(parts[0]), d))
    print(f"{len(samples)} samples")
    for depth, n, t0, t1 in legs:
        ss = [d for t, d in samples if t0 <= t <= t1]

        def stat(key, scale):
            v = [float(d[key]) * scale for d in ss if key in d and d[key] != "ERR"]
            return f"{min(v):.0f}/{sum(v) / len(v):.0f}/{max(v):.0f}" if v else "n/a"

        pw = "power1_average" if ss and "power1_average" in ss[0] else "power1_input"
        print(f"inflight {depth} x{n:3d}: {1e3 * (t1 - t0) / n:.3f} ms/step {B * n / (t1 - t0):6.0f} img/s | sclk MHz min/avg/max {stat('freq1_input', 1e-6)} | "
              f"W {stat(pw, 1e-6)} | temp {stat('temp1_input', 1e-3)} | {len(ss)} samples", flush=True)
