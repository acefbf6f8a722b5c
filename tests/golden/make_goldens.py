#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own modules
(read-only at /root/reference) on seeded synthetic inputs and OUR seeded weights.

Runs only in the build container (the reference never travels to the GPU box).  What is
committed is data: inputs, expected outputs, sampled elements — never reference source.

Reference modules imported: evals/models/ibot_transformers.py,
evals/models/probes.py, evals/utils/losses.py, evals/utils/optim.py — behind stub
``evals`` / ``evals.models`` / ``evals.utils`` packages because the shipped
``evals/models/__init__.py`` re-exports names that no longer exist.

    python tests/golden/make_goldens.py            # writes tests/golden/*.npz
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("MVP_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from oracle import probes as oprobes  # noqa: E402  (weight generators only)
from oracle import train as otrain  # noqa: E402
from oracle import vit as ovit  # noqa: E402


def _load_reference():
    sys.path.insert(0, REF)
    for name, sub in (("evals", "evals"), ("evals.models", "evals/models"), ("evals.utils", "evals/utils")):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, sub)]
        sys.modules[name] = m
    vt = importlib.import_module("evals.models.ibot_transformers")
    pr = importlib.import_module("evals.models.probes")
    ls = importlib.import_module("evals.utils.losses")
    op = importlib.import_module("evals.utils.optim")
    return vt, pr, ls, op


def _np(t):
    return t.detach().cpu().numpy()


def _ref_vit_taps(vt, model, images, layers, add_norm=True, patch=16):
    """Drive the reference VisionTransformer the way DINO.forward does (dino.py:164-210):
    the wrapper itself needs torchvision (absent), so its ~25 lines of glue are replayed
    here around the reference's own prepare_tokens / Block / nn.BatchNorm1d."""
    import torch.nn as nn
    import torch.nn.functional as F

    _, _, h, w = images.shape
    dh, dw = h % patch, w % patch
    if not (dh == 0 and dw == 0):
        ph, pw = patch - dh, patch - dw
        images = F.pad(images, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    h, w = images.shape[-2] // patch, images.shape[-1] // patch
    bns = nn.ModuleList([nn.BatchNorm1d(model.embed_dim) for _ in layers])
    x = model.prepare_tokens(images)
    tokens0 = x.clone()
    embeds = []
    for i, blk in enumerate(model.blocks):
        x = blk(x)
        if i in layers:
            if add_norm:
                embeds.append(bns[list(layers).index(i)](x.permute(0, 2, 1)).permute(0, 2, 1))
            else:
                embeds.append(x)
            if len(embeds) == len(layers):
                break
    outs = []
    for e in embeds:
        sp = e[:, -h * w:]
        outs.append(sp.reshape(sp.shape[0], h, w, -1).permute(0, 3, 1, 2).contiguous())
    running = [(bn.running_mean.clone(), bn.running_var.clone()) for bn in bns]
    return tokens0, outs, running


def golden_vit_tiny(vt, embed_dim=64, heads=4, fname="vit_tiny.npz"):
    """G1: tiny ViT, full tensors, non-square ragged input (pad + pos-embed interp).
    Two variants: 64/4 heads (head_dim 16, oracle only) and 128/2 heads (head_dim 64, the
    HIP attention kernel's head size)."""
    cfg = dict(embed_dim=embed_dim, depth=4, num_heads=heads, patch_size=16)
    sd = ovit.make_vit_weights(embed_dim=embed_dim, depth=4, seed=11)
    model = vt.VisionTransformer(qkv_bias=True, mlp_ratio=4, **cfg).eval()
    missing = model.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(5)
    out = {}
    for tag, shape in (("a", (3, 3, 64, 96)), ("b", (2, 3, 70, 100)), ("c", (2, 3, 224, 224))):
        images = torch.randn(*shape, generator=g)
        with torch.no_grad():
            tok, taps, running = _ref_vit_taps(vt, model, images, [0, 1, 2, 3])
            _, raw, _ = _ref_vit_taps(vt, model, images, [3], add_norm=False)
        out[f"{tag}_images"] = _np(images)
        out[f"{tag}_tokens0"] = _np(tok)
        for i, t in enumerate(taps):
            out[f"{tag}_tap{i}"] = _np(t)
            out[f"{tag}_rmean{i}"] = _np(running[i][0])
            out[f"{tag}_rvar{i}"] = _np(running[i][1])
        out[f"{tag}_raw_last"] = _np(raw[0])
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print("vit_tiny", {k: v.shape for k, v in out.items() if k.startswith("a_")})


def golden_vit_base(vt):
    """G2: ViT-B/16, weights regenerated from seed on the test side; store sampled
    elements + moments of every tap (full tensors would be ~19 MB per case)."""
    sd = ovit.make_vit_weights(seed=0)
    model = vt.vit_base(patch_size=16).eval()
    model.load_state_dict(sd, strict=True)
    out = {}
    for tag, (B, H, W), seed in (("b224", (2, 224, 224), 21), ("b480x640", (1, 480, 640), 22)):
        images = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(seed))
        with torch.no_grad():
            tok, taps, running = _ref_vit_taps(vt, model, images, [2, 5, 8, 11])
            _, raw, _ = _ref_vit_taps(vt, model, images, [2, 5, 8, 11], add_norm=False)
        gi = torch.Generator().manual_seed(99)
        out[f"{tag}_seed"] = np.array([seed, B, H, W])
        for i, (t, r) in enumerate(zip(taps, raw)):
            idx = torch.randint(0, t.numel(), (4096,), generator=gi)
            out[f"{tag}_idx{i}"] = _np(idx)
            out[f"{tag}_tap{i}_samples"] = _np(t.flatten()[idx])
            out[f"{tag}_raw{i}_samples"] = _np(r.flatten()[idx])
            out[f"{tag}_tap{i}_moments"] = np.array([t.mean().item(), t.std().item(), t.abs().max().item(), t.norm().item()])
            out[f"{tag}_raw{i}_moments"] = np.array([r.mean().item(), r.std().item(), r.abs().max().item(), r.norm().item()])
            out[f"{tag}_rmean{i}"] = _np(running[i][0])
            out[f"{tag}_rvar{i}"] = _np(running[i][1])
        print("vit_base", tag, [tuple(t.shape) for t in taps])
    np.savez_compressed(os.path.join(OUT, "vit_base.npz"), **out)


def _grads(module):
    return {n: _np(p.grad) for n, p in module.named_parameters()}


def golden_probes(pr):
    """G3: every DepthHead / SurfaceNormalHead variant on tiny dims: outputs + param grads."""
    out = {}
    g = torch.Generator().manual_seed(3)
    B, C, h, w = 2, 24, 5, 7
    vit_feats = [torch.randn(B, C, h, w, generator=g) for _ in range(4)]
    out["vit_feats"] = np.stack([_np(f) for f in vit_feats])
    # ResNet-style pyramid (taps at 8x,4x,2x,1x of the coarsest map, DPT bilinear x2 chain)
    rdims = [(8, 0), (12, 0), (16, 0), (20, 0)]
    res_feats = [torch.randn(B, rdims[i][0], 3 * 2 ** (3 - i), 4 * 2 ** (3 - i), generator=g) for i in range(4)]
    for i, f in enumerate(res_feats):
        out[f"res_feat{i}"] = _np(f)

    cases = []
    for k in (1, 3):
        for pt in ("bindepth", "sigdepth"):
            cases.append((f"depth_linear_k{k}_{pt}", "depth", dict(head_type="linear", kernel_size=k, prediction_type=pt), "vit"))
    cases.append(("depth_dpt_k3_bindepth", "depth", dict(head_type="dpt", kernel_size=3, prediction_type="bindepth", hidden_dim=16), "vit"))
    cases.append(("depth_dpt_k3_sigdepth_res", "depth", dict(head_type="dpt", kernel_size=3, prediction_type="sigdepth", hidden_dim=16), "res"))
    cases.append(("snorm_linear_k1_ua", "snorm", dict(head_type="linear", kernel_size=1, uncertainty_aware=True), "vit"))
    cases.append(("snorm_dpt_k3_ua", "snorm", dict(head_type="dpt", kernel_size=3, uncertainty_aware=True, hidden_dim=16), "vit"))
    cases.append(("snorm_dpt_k3_res", "snorm", dict(head_type="dpt", kernel_size=3, uncertainty_aware=False, hidden_dim=16), "res"))

    for name, kind, kw, src in cases:
        feat_dim = [C] * 4 if src == "vit" else [tuple(d) for d in rdims]
        feats = vit_feats if src == "vit" else res_feats
        if kind == "depth":
            probe = pr.DepthHead(feat_dim=feat_dim, min_depth=0.001, max_depth=10, **kw)
        else:
            probe = pr.SurfaceNormalHead(feat_dim=feat_dim, **kw)
        # deterministic weights: ours, through the reference key layout
        gen = oprobes.make_linear_head_weights if kw["head_type"] == "linear" else None
        odim = probe.head.conv.out_channels if kw["head_type"] == "linear" else probe.head.out_conv[2].out_channels
        if kw["head_type"] == "linear":
            sd = oprobes.make_linear_head_weights([C] * 4, odim, kw["kernel_size"], seed=17)
        else:
            sd = oprobes.make_dpt_weights(feat_dim, odim, hidden=16, k=kw["kernel_size"], seed=17)
        probe.load_state_dict(sd, strict=True)
        y = probe([f.clone() for f in feats])
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(23))
        (y * gy).sum().backward()
        out[f"{name}__out"] = _np(y)
        out[f"{name}__gy"] = _np(gy)
        out[f"{name}__name"] = np.array(probe.name)
        for n, gr in _grads(probe).items():
            out[f"{name}__grad__{n}"] = gr
        print("probe", name, tuple(y.shape), probe.name)
    np.savez_compressed(os.path.join(OUT, "probes.npz"), **out)


def golden_probes_multiscale(pr):
    """G3b: MultiscaleHead (probes.py:435-458; DepthHead's default head_type) on tiny dims, ViT-style equal-resolution taps and a
    ResNet-style pyramid (maps resampled to the last one's size): outputs + parameter gradients of the reference module."""
    out = {}
    g = torch.Generator().manual_seed(33)
    B, C, h, w = 2, 24, 5, 7
    vit_feats = [torch.randn(B, C, h, w, generator=g) for _ in range(4)]
    rdims = [(8, 0), (12, 0), (16, 0), (20, 0)]
    res_feats = [torch.randn(B, rdims[i][0], 3 * 2 ** (3 - i), 4 * 2 ** (3 - i), generator=g) for i in range(4)]
    out["vit_feats"] = np.stack([_np(f) for f in vit_feats])
    for i, f in enumerate(res_feats):
        out[f"res_feat{i}"] = _np(f)
    for name, kind, kw, src in (("depth_ms_bindepth", "depth", dict(prediction_type="bindepth"), "vit"),
                                ("depth_ms_sigdepth_res", "depth", dict(prediction_type="sigdepth"), "res"),
                                ("snorm_ms_ua", "snorm", dict(uncertainty_aware=True), "vit")):
        feat_dim = [C] * 4 if src == "vit" else [d[0] for d in rdims]  # (the reference's MultiscaleHead takes plain ints only)
        feats = vit_feats if src == "vit" else res_feats
        probe = (pr.DepthHead if kind == "depth" else pr.SurfaceNormalHead)(feat_dim=feat_dim, head_type="multiscale", hidden_dim=16, kernel_size=1, **kw)
        odim = probe.head.conv_out[2].out_channels
        probe.load_state_dict(oprobes.make_multiscale_weights(feat_dim, odim, hidden=16, k=1, seed=19), strict=True)
        y = probe([f.clone() for f in feats])
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(29))
        (y * gy).sum().backward()
        out[f"{name}__out"], out[f"{name}__gy"], out[f"{name}__name"] = _np(y), _np(gy), np.array(probe.name)
        for n, gr in _grads(probe).items():
            out[f"{name}__grad__{n}"] = gr
        print("probe", name, tuple(y.shape), probe.name)
    # kernel_size 3: make_conv builds UN-PADDED convs (probes.py:400-412), every conv shrinks its map by 2
    k3_feats = [torch.randn(B, C, 6, 7, generator=g) for _ in range(4)]
    out["k3_feats"] = np.stack([_np(f) for f in k3_feats])
    probe = pr.DepthHead(feat_dim=[C] * 4, head_type="multiscale", hidden_dim=16, kernel_size=3, prediction_type="sigdepth")
    probe.load_state_dict(oprobes.make_multiscale_weights([C] * 4, 1, hidden=16, k=3, seed=23), strict=True)
    y = probe([f.clone() for f in k3_feats])
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(31))
    (y * gy).sum().backward()
    out["depth_ms_k3__out"], out["depth_ms_k3__gy"], out["depth_ms_k3__name"] = _np(y), _np(gy), np.array(probe.name)
    for n, gr in _grads(probe).items():
        out[f"depth_ms_k3__grad__{n}"] = gr
    print("probe depth_ms_k3", tuple(y.shape), probe.name)
    np.savez_compressed(os.path.join(OUT, "probes_multiscale.npz"), **out)


def golden_losses(ls):
    """G4: DepthLoss (B in {1,2,3,5,8,16} pins quirk Q1), sig_loss, gradient_loss,
    angular_loss (UA on/off): values + input grads."""
    out = {}
    for B in (1, 2, 3, 5, 8, 16):
        g = torch.Generator().manual_seed(100 + B)
        pred = (torch.rand(B, 1, 12, 10, generator=g) * 9 + 0.01).requires_grad_(True)
        tgt = torch.rand(B, 1, 12, 10, generator=g) * 12  # some > max_depth=10 -> zeroed
        tgt[torch.rand(B, 1, 12, 10, generator=g) < 0.15] = 0
        t_in = tgt.clone()
        loss = ls.DepthLoss()(pred, tgt)
        loss.backward()
        out[f"depth_B{B}_pred"] = _np(pred)
        out[f"depth_B{B}_target"] = _np(t_in)
        out[f"depth_B{B}_target_after"] = _np(tgt)
        out[f"depth_B{B}_loss"] = np.array(loss.item())
        out[f"depth_B{B}_grad"] = _np(pred.grad)
        with torch.no_grad():
            out[f"depth_B{B}_sig"] = np.array(ls.sig_loss(pred, tgt).item())
            out[f"depth_B{B}_gradloss"] = np.array(float(ls.gradient_loss(pred, tgt)))
    for ua in (False, True):
        g = torch.Generator().manual_seed(200 + int(ua))
        C = 4 if ua else 3
        pred = torch.randn(3, C, 9, 11, generator=g).requires_grad_(True)
        gt = torch.randn(3, 3, 9, 11, generator=g)
        gt = gt / gt.norm(dim=1, keepdim=True)
        mask = torch.rand(3, 1, 9, 11, generator=g) > 0.2
        loss = ls.angular_loss(pred, gt, mask, uncertainty_aware=ua)
        loss.backward()
        tag = f"ang_ua{int(ua)}"
        out[f"{tag}_pred"], out[f"{tag}_gt"], out[f"{tag}_mask"] = _np(pred), _np(gt), _np(mask)
        out[f"{tag}_loss"], out[f"{tag}_grad"] = np.array(loss.item()), _np(pred.grad)
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **out)
    print("losses", len(out))


def golden_optim(op, pr, ls):
    """G5 schedule table + G6 5-step AdamW trajectory of a linear bindepth probe."""
    import torch.nn.functional as F

    out = {}
    steps = np.arange(0, 101)
    out["sched_steps"] = steps
    out["sched_vals"] = np.array([op.cosine_decay_linear_warmup(int(s), 100, 15) for s in steps], dtype=np.float64)
    out["sched_vals_frac"] = np.array([op.cosine_decay_linear_warmup(int(s), 10 * 7, 1.5 * 7) for s in range(0, 70)], dtype=np.float64)

    C, B, h, w = 16, 4, 4, 5
    probe = pr.DepthHead(feat_dim=[C] * 4, head_type="linear", kernel_size=1, prediction_type="bindepth", min_depth=0.001, max_depth=10)
    probe.load_state_dict(oprobes.make_linear_head_weights([C] * 4, 256, 1, seed=5), strict=True)
    optimizer = torch.optim.AdamW([{"params": probe.parameters(), "lr": 5e-4}])
    sched = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda e: op.cosine_decay_linear_warmup(e, 40, 3))
    loss_fn = ls.DepthLoss()
    traj, lrs = [], []
    for s in range(5):
        g = torch.Generator().manual_seed(300 + s)
        feats = [torch.randn(B, C, h, w, generator=g) for _ in range(4)]
        tgt = torch.rand(B, 1, 4 * h + 3, 4 * w + 2, generator=g) * 9.9 + 0.05
        tgt[torch.rand(tgt.shape, generator=g) < 0.1] = 0
        out[f"traj_feats{s}"] = np.stack([_np(f) for f in feats])
        out[f"traj_target{s}"] = _np(tgt)
        optimizer.zero_grad()
        pred = F.interpolate(probe(feats), size=tgt.shape[-2:], mode="bilinear")
        loss = loss_fn(pred, tgt)
        loss.backward()
        lrs.append(optimizer.param_groups[0]["lr"])
        optimizer.step()
        sched.step()
        traj.append(loss.item())
    out["traj_losses"] = np.array(traj)
    out["traj_lrs"] = np.array(lrs)
    out["traj_final_weight"] = _np(probe.head.conv.weight)
    out["traj_final_bias"] = _np(probe.head.conv.bias)
    np.savez_compressed(os.path.join(OUT, "optim.npz"), **out)
    print("optim", traj, lrs)


def golden_step(vt, pr, ls, op):
    """End-to-end: tiny ViT -> 4 taps (train-mode BN) -> linear bindepth probe -> bilinear
    -> DepthLoss -> backward -> AdamW, 3 steps, exactly the train_depth.py loop body."""
    import torch.nn.functional as F

    D, depth = 128, 4  # 2 heads of 64 = the HIP attention kernel's head size
    sd = ovit.make_vit_weights(embed_dim=D, depth=depth, seed=31)
    model = vt.VisionTransformer(qkv_bias=True, mlp_ratio=4, embed_dim=D, depth=depth, num_heads=2, patch_size=16).eval()
    model.load_state_dict(sd, strict=True)
    probe = pr.DepthHead(feat_dim=[D] * 4, head_type="linear", kernel_size=1, prediction_type="bindepth")
    probe.load_state_dict(oprobes.make_linear_head_weights([D] * 4, 256, 1, seed=32), strict=True)
    optimizer = torch.optim.AdamW([{"params": probe.parameters(), "lr": 5e-4}])
    sched = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda e: op.cosine_decay_linear_warmup(e, 30, 2))
    loss_fn = ls.DepthLoss()
    out = {}
    losses = []
    for s in range(3):
        images, tgt = otrain.synthetic_depth_batch(4, 64, 80, rank=0, step=s)
        optimizer.zero_grad()
        with torch.no_grad():
            _, feats, _ = _ref_vit_taps(vt, model, images, [0, 1, 2, 3])
            feats = [f.detach() for f in feats]
        pred = F.interpolate(probe(feats), size=tgt.shape[-2:], mode="bilinear")
        loss = loss_fn(pred, tgt)
        loss.backward()
        if s == 0:
            out["grad_w0"] = _np(probe.head.conv.weight.grad)
            out["grad_b0"] = _np(probe.head.conv.bias.grad)
            out["pred0"] = _np(pred)
        optimizer.step()
        sched.step()
        losses.append(loss.item())
    out["losses"] = np.array(losses)
    out["final_weight"] = _np(probe.head.conv.weight)
    out["final_bias"] = _np(probe.head.conv.bias)
    np.savez_compressed(os.path.join(OUT, "step_tiny.npz"), **out)
    print("step_tiny", losses)


def golden_metrics():
    """G7: evaluate_depth (global metrics, scale-aware and scale-invariant), match_scale_and_shift,
    evaluate_surface_norm (global) from evals/utils/metrics.py.  That module needs ``loguru`` (absent):
    a 3-line stub logger is registered first."""
    lg = types.ModuleType("loguru")

    class _L:
        def warning(self, *a, **k):
            pass

        info = warning

    lg.logger = _L()
    sys.modules.setdefault("loguru", lg)
    mt = importlib.import_module("evals.utils.metrics")
    out = {}
    g = torch.Generator().manual_seed(400)
    B, H, W = 5, 37, 41
    pr = torch.rand(B, 1, H, W, generator=g) * 9 + 0.05
    gt = torch.rand(B, 1, H, W, generator=g) * 9 + 0.05
    gt[torch.rand(gt.shape, generator=g) < 0.2] = 0
    gt[4] = 0  # an image without any valid pixel (num_valid -> 1e-6 path)
    seg = torch.zeros(B, H, W, dtype=torch.long)
    out["pred"], out["gt"] = _np(pr), _np(gt)
    for tag, si in (("sa", False), ("si", True)):
        gm = mt.evaluate_depth(pr, gt, seg, scale_invariant=si, is_navi=True)[0]
        for k, v in gm.items():
            out[f"{tag}_{k}"] = _np(v.reshape(B))
    out["matched"] = _np(mt.match_scale_and_shift(pr, gt))
    sn = torch.randn(B, 4, H, W, generator=g)
    sg = torch.randn(B, 3, H, W, generator=g)
    sg = sg / sg.norm(dim=1, keepdim=True)
    sg[:, :, :5] = 0  # invalid rows (|gt| sum == 0)
    out["sn_pred"], out["sn_gt"] = _np(sn), _np(sg)
    sm = mt.evaluate_surface_norm(sn, sg, seg, is_navi=True)[0]
    for k, v in sm.items():
        out[f"sn_{k}"] = _np(v.reshape(B))
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **out)
    print("metrics", {k: v.shape for k, v in out.items() if k.startswith("sa_")})


def _stub_loguru():
    lg = types.ModuleType("loguru")

    class _L:
        def warning(self, *a, **k):
            pass

        info = warning

    lg.logger = _L()
    sys.modules.setdefault("loguru", lg)


def _seg_rows(seg_metrics):
    return np.array([[m["segment_id"], m["image_idx"], m["area"], m["d1_ratio"]] for m in seg_metrics], dtype=np.float64)


def golden_metrics_breakdown():
    """G7b: the per-level ("centroid level"), stuff/things and per-segment parts of evaluate_depth (metrics.py:179-358) and
    evaluate_surface_norm (metrics.py:441-577) on a synthetic INTEGER segmentation map (OneFormer ADE20K ids 0..149; 11, 17, 40
    and 68 are neither STUFF nor THINGS in evals/utils/oneformer_id2label.py).  Two geometries: H divisible by num_levels and
    not; one image without any valid pixel; one segment id that only occurs on invalid pixels."""
    _stub_loguru()
    mt = importlib.import_module("evals.utils.metrics")
    out = {}
    for tag, (B, H, W), seed in (("a", (3, 40, 50), 410), ("b", (2, 37, 31), 411)):
        g = torch.Generator().manual_seed(seed)
        pr = torch.rand(B, 1, H, W, generator=g) * 9 + 0.05
        gt = pr * (1 + 0.35 * torch.randn(B, 1, H, W, generator=g)).clamp(min=0.2)  # d1 ~ 0.5: both sides of every threshold
        gt[torch.rand(gt.shape, generator=g) < 0.2] = 0
        ids = torch.tensor([0, 3, 5, 7, 8, 11, 17, 26, 40, 68, 100, 128, 149])
        seg = ids[torch.randint(0, len(ids), (B, H // 4 + 1, W // 4 + 1), generator=g)]
        seg = seg.repeat_interleave(4, 1).repeat_interleave(4, 2)[:, :H, :W].contiguous()  # 4x4 blobs
        gt[:, 0][seg == 17] = 0        # id 17 only on invalid pixels: still listed by torch.unique
        if tag == "a":
            gt[2] = 0                  # an image without valid pixels
        sn = torch.randn(B, 4, H, W, generator=g)
        sg = torch.randn(B, 3, H, W, generator=g)
        sg = sg / sg.norm(dim=1, keepdim=True)
        sn[:, :3] = sn[:, :3] * 0.6 + sg * 1.2                       # errors spread around the 11.25/22.5/30 degree thresholds
        sg[:, :, :5] = 0
        out[f"{tag}_pred"], out[f"{tag}_gt"], out[f"{tag}_seg"] = _np(pr), _np(gt), _np(seg).astype(np.int32)
        out[f"{tag}_sn_pred"], out[f"{tag}_sn_gt"] = _np(sn), _np(sg)
        for mode, si in (("sa", False), ("si", True)):
            gm, lv, sm = mt.evaluate_depth(pr, gt, seg, scale_invariant=si, is_navi=False)
            for k, v in gm.items():
                out[f"{tag}_{mode}_{k}"] = _np(v.reshape(B))
            for L, d in lv.items():
                for k, v in d.items():
                    out[f"{tag}_{mode}_{L}_{k}"] = _np(v.reshape(B))
            out[f"{tag}_{mode}_segments"] = _seg_rows(sm)
        gm, lv, sm = mt.evaluate_depth(pr, gt, seg, image_average=True, num_levels=3, is_navi=False)
        out[f"{tag}_avg3_rmse"] = _np(gm["rmse"])
        for L, d in lv.items():
            out[f"{tag}_avg3_{L}_d1"] = _np(d["d1"])
        gm, lv, sm = mt.evaluate_surface_norm(sn, sg, seg, is_navi=False)
        for k, v in gm.items():
            out[f"{tag}_sn_{k}"] = _np(v.reshape(B))
        for L, d in lv.items():
            for k, v in d.items():
                out[f"{tag}_sn_{L}_{k}"] = _np(v.reshape(B))
        out[f"{tag}_sn_segments"] = _seg_rows(sm)
    np.savez_compressed(os.path.join(OUT, "metrics_seg.npz"), **out)
    print("metrics_seg", len(out), "arrays;", {k: v.shape for k, v in out.items() if k.startswith("a_sa_level_1")})


def golden_si_train(ls):
    """The scale_invariant branch of the train loop (train_depth.py:114-118): pred -> match_scale_and_shift (scale / shift
    DETACHED, metrics.py:775-776) -> clamp(0.001, 1.0) -> DepthLoss; loss and d loss / d pred from the reference's own functions."""
    _stub_loguru()
    mt = importlib.import_module("evals.utils.metrics")
    out = {}
    g = torch.Generator().manual_seed(600)
    B, H, W = 3, 24, 32
    tgt = torch.rand(B, 1, H, W, generator=g) * 0.9 + 0.05           # relative depth in (0, 1] as navi_reldepth
    tgt[torch.rand(tgt.shape, generator=g) < 0.15] = 0
    pred = (tgt * 1.7 + 0.2 + 0.25 * torch.randn(B, 1, H, W, generator=g)).requires_grad_(True)  # some values leave [0.001, 1] after the fit
    p2 = mt.match_scale_and_shift(pred, tgt)
    p3 = p2.clamp(min=0.001, max=1.0)
    loss = ls.DepthLoss()(p3, tgt.clone())
    loss.backward()
    out.update(pred=_np(pred), target=_np(tgt), matched=_np(p2), clamped=_np(p3), loss=np.array(loss.item()), grad=_np(pred.grad))
    np.savez_compressed(os.path.join(OUT, "si_train.npz"), **out)
    print("si_train loss", loss.item(), "clamped fraction", float(((p2 < 0.001) | (p2 > 1.0)).float().mean()))


def golden_spair():
    """G8: argmax_2d (evals/utils/correspondence.py:179-190) from the reference module itself.  Its top-level
    ``import faiss`` / ``faiss.contrib.torch_utils`` / ``faiss.StandardGpuResources()`` (correspondence.py:4-5,11) are
    satisfied by an empty stub (faiss is not installed; argmax_2d does not use it).  Heat-maps are built the way
    compute_errors does (evaluate_spair_correspondence.py:59-83: torch core ops + einops) and include exact ties."""
    fs = types.ModuleType("faiss")
    fs.StandardGpuResources = lambda: None
    fc = types.ModuleType("faiss.contrib")
    ft = types.ModuleType("faiss.contrib.torch_utils")
    fs.contrib, fc.torch_utils = fc, ft
    for n, m in (("faiss", fs), ("faiss.contrib", fc), ("faiss.contrib.torch_utils", ft)):
        sys.modules.setdefault(n, m)
    co = importlib.import_module("evals.utils.correspondence")
    import torch.nn.functional as F
    from einops import einsum

    out = {}
    g = torch.Generator().manual_seed(500)
    C, h, w, K = 96, 13, 17, 11                               # non-square map: (col,row) order matters
    feats = torch.randn(2, C, h, w, generator=g)
    kps = torch.rand(K, 2, generator=g)
    kps[0] = torch.tensor([0.0, 0.0]); kps[1] = torch.tensor([1.0, 1.0])
    kps[2] = torch.tensor([5 / (w - 1), 7 / (h - 1)])          # exactly on source pixel (5, 7)
    # exact ties: the target carries the descriptor of keypoint 2 at two pixels (rows 3 and 9): argmax must take the first
    fn = F.normalize(feats, p=2, dim=1)
    feats[1] *= 0.05
    feats[1][:, 9, 4] = fn[0][:, 7, 5] * 3.0
    feats[1][:, 3, 12] = fn[0][:, 7, 5] * 3.0
    fn = F.normalize(feats, p=2, dim=1)
    ndc = (kps * 2 - 1)[None, None]
    kf = F.grid_sample(fn[0][None], ndc, mode="bilinear", align_corners=True)[0, :, 0].t()
    heat = einsum(kf, fn[1], "k f, f h w -> k h w")
    out["feats"], out["kps01"], out["heat"] = _np(feats), _np(kps), _np(heat)
    out["pred_max"] = _np(co.argmax_2d(heat, max_value=True))
    out["pred_min"] = _np(co.argmax_2d(heat, max_value=False))
    # integer-valued maps with many ties (plateaus), incl. a constant map and maxima in the last row / column
    ti = torch.randint(-3, 4, (9, 6, 10), generator=g).float()
    ti[0] = 2.0
    ti[1] = -1.0; ti[1, 5, 9] = 7.0
    ti[2] = 1.0; ti[2, 0, 0] = -7.0
    out["tie_heat"] = _np(ti)
    out["tie_max"] = _np(co.argmax_2d(ti, max_value=True))
    out["tie_min"] = _np(co.argmax_2d(ti, max_value=False))
    np.savez_compressed(os.path.join(OUT, "spair.npz"), **out)
    print("spair", out["pred_max"][:4].tolist(), out["tie_max"][:3].tolist(), out["tie_min"][:3].tolist())


def golden_mae():
    """Pins the MAE path (evals/models/mae.py:33,91-104,203-237) to the third-party arithmetic it calls: HF transformers' ViTMAE encoder.

    ``transformers.ViTMAEModel(ViTMAEConfig(...))`` is built from a config (no fetch) with seeded weights, and the reference wrapper's
    glue is replayed around ITS modules: embed_forward without masking (patch embeddings + pos[1:], CLS + pos[0]: mae.py:91-104), the
    encoder layers one by one collecting hidden_states (index 0 = the embedding output, index i = output of layer i-1, as
    ``output_hidden_states=True`` returns them), taps hidden_states[multilayers] -> nn.BatchNorm1d (train mode) -> dense map
    (mae.py:216-236).  Caveat, stated here because the fixture cannot: the image holds transformers 5.15 (layers.i.attention.q_proj
    ...), the reference pins 4.29.2 (encoder.layer.i.attention.attention.query ...): same pre-LN block (LayerNorm eps 1e-12, separate
    q / k / v linears with bias, exact-erf GELU), different attribute names; the weights are saved under the 4.29.2 names the wrapper's
    checkpoint loader expects.  The model's own position embeddings (HF's 2-D sincos initialisation) are stored as well: they pin the
    sincos table the wrapper rebuilds in resize_pos_embed (mae.py:74-89, utils.py:75-102)."""
    import torch.nn as nn
    from transformers import ViTMAEConfig, ViTMAEModel

    torch.manual_seed(1234)
    D, depth, heads, img = 128, 4, 2, 96
    cfg = ViTMAEConfig(hidden_size=D, num_hidden_layers=depth, num_attention_heads=heads, intermediate_size=4 * D, image_size=img, patch_size=16, mask_ratio=0.0)
    m = ViTMAEModel(cfg).eval()
    m.embeddings.patch_embeddings.projection._is_hf_initialized = False  # (post_init has marked the module done: run the sincos initialisation again)
    m.embeddings.initialize_weights()  # HF's own 2-D sincos table for the config's grid (what from_pretrained checkpoints carry)
    with torch.no_grad():  # HF initialises linears with std 0.02 and zero biases: give every tensor some signal
        for n, p_ in m.named_parameters():
            if "position_embeddings" in n:
                continue
            if p_.dim() == 1:
                p_.copy_(torch.randn_like(p_) * 0.05 + (1.0 if "layernorm" in n and n.endswith("weight") else 0.0))
            elif "cls_token" in n:
                p_.copy_(torch.randn_like(p_) * 0.02)
    images = torch.randn(2, 3, img, img, generator=torch.Generator().manual_seed(77))
    multilayers = [depth // 4 - 1, depth // 2 - 1, depth // 4 * 3 - 1, depth - 1]
    with torch.no_grad():
        emb = m.embeddings.patch_embeddings(images) + m.embeddings.position_embeddings[:, 1:, :]
        cls = (m.embeddings.cls_token + m.embeddings.position_embeddings[:, :1, :]).expand(emb.shape[0], -1, -1)
        x = torch.cat((cls, emb), dim=1)
        hidden = [x]
        for layer in m.layers:
            x = layer(x)
            hidden.append(x)
        gh = gw = img // 16
        out = {"images": _np(images), "multilayers": np.asarray(multilayers), "heads": np.asarray(heads), "pos_embed_hf": _np(m.embeddings.position_embeddings)}
        for j, li in enumerate(multilayers):
            bn = nn.BatchNorm1d(D)  # fresh, train mode, identity affine: as the wrapper's batchnorms at their first step
            xi = hidden[li]
            y = bn(xi.permute(0, 2, 1)).permute(0, 2, 1)
            out[f"tokens_{j}"] = _np(xi)
            out[f"dense_{j}"] = _np(y[:, 1:].reshape(2, gh, gw, D).permute(0, 3, 1, 2).contiguous())
    sd = m.state_dict()
    ren = {"embeddings.cls_token": "embeddings.cls_token", "embeddings.position_embeddings": "embeddings.position_embeddings",
           "embeddings.patch_embeddings.projection.weight": "embeddings.patch_embeddings.projection.weight",
           "embeddings.patch_embeddings.projection.bias": "embeddings.patch_embeddings.projection.bias",
           "layernorm.weight": "layernorm.weight", "layernorm.bias": "layernorm.bias"}
    sub = {"attention.q_proj": "attention.attention.query", "attention.k_proj": "attention.attention.key", "attention.v_proj": "attention.attention.value",
           "attention.o_proj": "attention.output.dense", "layernorm_before": "layernorm_before", "layernorm_after": "layernorm_after",
           "mlp.fc1": "intermediate.dense", "mlp.fc2": "output.dense"}
    for k, v in sd.items():
        if k in ren:
            out["w:" + ren[k]] = _np(v)
            continue
        assert k.startswith("layers."), k
        _, i, rest = k.split(".", 2)
        mod, leaf = rest.rsplit(".", 1)
        out[f"w:encoder.layer.{i}.{sub[mod]}.{leaf}"] = _np(v)
    np.savez_compressed(os.path.join(OUT, "mae_tiny.npz"), **out)
    print("mae_tiny", [float(np.abs(out[f"dense_{j}"]).mean()) for j in range(4)], "pos_embed |max|", float(np.abs(out["pos_embed_hf"]).max()))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    vt, pr, ls, op = _load_reference()
    which = sys.argv[1:] or ["vit_tiny", "vit_tiny128", "vit_base", "probes", "losses", "optim", "step", "metrics", "metrics_seg", "spair", "probes_multiscale", "si_train", "mae"]
    if "vit_tiny" in which:
        golden_vit_tiny(vt)
    if "vit_tiny128" in which:
        golden_vit_tiny(vt, embed_dim=128, heads=2, fname="vit_tiny128.npz")
    if "probes" in which:
        golden_probes(pr)
    if "probes_multiscale" in which:
        golden_probes_multiscale(pr)
    if "losses" in which:
        golden_losses(ls)
    if "optim" in which:
        golden_optim(op, pr, ls)
    if "step" in which:
        golden_step(vt, pr, ls, op)
    if "vit_base" in which:
        golden_vit_base(vt)
    if "metrics" in which:
        golden_metrics()
    if "metrics_seg" in which:
        golden_metrics_breakdown()
    if "si_train" in which:
        golden_si_train(ls)
    if "spair" in which:
        golden_spair()
    if "mae" in which:
        golden_mae()


if __name__ == "__main__":
    main()
