"""Pin the CPU oracle to the golden vectors produced by the reference's own modules
(tests/golden/make_goldens.py).  CPU only; these run with -m "not gpu"."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, max_rel, rel_l2
from oracle import losses as olosses
from oracle import optim as ooptim
from oracle import probes as oprobes
from oracle import train as otrain
from oracle import vit as ovit

TOL = 2e-5  # fp32 CPU restatement vs fp32 CPU reference: op order differs only inside torch kernels


def T(a):
    return torch.from_numpy(np.array(a))


# ------------------------------------------------------------------ ViT
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_vit_tiny_full(tag):
    g = load_golden("vit_tiny.npz")
    sd = ovit.make_vit_weights(embed_dim=64, depth=4, seed=11)
    images = T(g[f"{tag}_images"])
    tok = ovit.prepare_tokens(sd, ovit.center_padding(images, 16))
    assert rel_l2(tok.numpy(), g[f"{tag}_tokens0"]) < TOL
    running = [(torch.zeros(64), torch.ones(64)) for _ in range(4)]
    taps = ovit.vit_dense_features(sd, images, [0, 1, 2, 3], heads=4, bn_running=running)
    for i, t in enumerate(taps):
        assert t.shape == g[f"{tag}_tap{i}"].shape
        assert rel_l2(t.numpy(), g[f"{tag}_tap{i}"]) < TOL, (tag, i)
        assert rel_l2(running[i][0].numpy(), g[f"{tag}_rmean{i}"]) < TOL
        assert rel_l2(running[i][1].numpy(), g[f"{tag}_rvar{i}"]) < TOL
    raw = ovit.vit_dense_features(sd, images, [3], heads=4, add_norm=False)
    assert rel_l2(raw.numpy(), g[f"{tag}_raw_last"]) < TOL


def test_vit_base_224_samples():
    g = load_golden("vit_base.npz")
    seed, B, H, W = [int(v) for v in g["b224_seed"]]
    sd = ovit.make_vit_weights(seed=0)
    images = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(seed))
    taps = ovit.vit_dense_features(sd, images, [2, 5, 8, 11])
    raw = ovit.vit_dense_features(sd, images, [2, 5, 8, 11], add_norm=False)
    for i in range(4):
        idx = T(g[f"b224_idx{i}"])
        assert rel_l2(taps[i].flatten()[idx].numpy(), g[f"b224_tap{i}_samples"]) < 1e-4
        assert rel_l2(raw[i].flatten()[idx].numpy(), g[f"b224_raw{i}_samples"]) < 1e-4
        m = g[f"b224_tap{i}_moments"]
        assert abs(taps[i].norm().item() - m[3]) / m[3] < 1e-5


def _mae_golden_fused_weights(g):
    """The golden's HF ViT-MAE weights (stored under the transformers 4.29.2 key names) in the fused DINO-style layout."""
    from mvp import backbone as bb

    return bb.hf_vitmae_to_fused({k[2:]: T(g[k]) for k in g.files if k.startswith("w:")})


def test_mae_vs_hf_vitmae_golden():
    """oracle/vit.py with the MAE settings (LayerNorm eps 1e-12, taps = block INPUTS: HF hidden_states indexing, quirk Q4; fixed
    sincos pos-embed) against transformers' own ViTMAE encoder layers driven the way evals/models/mae.py:91-104,203-237 drives them
    (golden mae_tiny.npz; generator: tests/golden/make_goldens.py::golden_mae, which states the 5.15-vs-4.29.2 caveat).  Also pins
    the sincos table the wrapper rebuilds in resize_pos_embed to HF's own initialisation."""
    g = load_golden("mae_tiny.npz")
    sd = _mae_golden_fused_weights(g)
    heads, layers = int(g["heads"]), [int(v) for v in g["multilayers"]]
    images = T(g["images"])
    assert np.abs(ovit.sincos_pos_embed_2d(128, (6, 6), True).numpy() - g["pos_embed_hf"][0]).max() < 1e-6
    toks = ovit.vit_dense_features(sd, images, layers, heads=heads, add_norm=False, ln_eps=1e-12, pos_mode="fixed", tap_input_of_block=True, return_tokens=True)
    dense = ovit.vit_dense_features(sd, images, layers, heads=heads, ln_eps=1e-12, pos_mode="fixed", tap_input_of_block=True)
    for j in range(4):
        assert rel_l2(toks[j].numpy(), g[f"tokens_{j}"]) < TOL, j
        assert rel_l2(dense[j].numpy(), g[f"dense_{j}"]) < TOL, j


def test_center_padding_quirk():
    x = torch.ones(1, 3, 32, 35)
    y = ovit.center_padding(x, 16)
    # ragged W only: H still receives a full patch of padding (reference quirk)
    assert y.shape == (1, 3, 48, 48)
    assert ovit.center_padding(torch.ones(1, 3, 32, 48), 16).shape == (1, 3, 32, 48)


# ------------------------------------------------------------------ probes
def _probe_cases():
    return [
        ("depth_linear_k1_bindepth", "depth", "linear", 1, "bindepth", "vit"),
        ("depth_linear_k1_sigdepth", "depth", "linear", 1, "sigdepth", "vit"),
        ("depth_linear_k3_bindepth", "depth", "linear", 3, "bindepth", "vit"),
        ("depth_linear_k3_sigdepth", "depth", "linear", 3, "sigdepth", "vit"),
        ("depth_dpt_k3_bindepth", "depth", "dpt", 3, "bindepth", "vit"),
        ("depth_dpt_k3_sigdepth_res", "depth", "dpt", 3, "sigdepth", "res"),
        ("snorm_linear_k1_ua", "snorm", "linear", 1, 4, "vit"),
        ("snorm_dpt_k3_ua", "snorm", "dpt", 3, 4, "vit"),
        ("snorm_dpt_k3_res", "snorm", "dpt", 3, 3, "res"),
    ]


@pytest.mark.parametrize("case", _probe_cases(), ids=lambda c: c[0])
def test_probe_fwd_bwd(case):
    name, kind, head, k, pt, src = case
    g = load_golden("probes.npz")
    C = 24
    rdims = [(8, 0), (12, 0), (16, 0), (20, 0)]
    if src == "vit":
        feats = [T(g["vit_feats"][i]) for i in range(4)]
        fdim = [C] * 4
    else:
        feats = [T(g[f"res_feat{i}"]) for i in range(4)]
        fdim = rdims
    odim = (256 if pt == "bindepth" else 1) if kind == "depth" else pt
    if head == "linear":
        sd = oprobes.make_linear_head_weights([C] * 4, odim, k, seed=17)
    else:
        sd = oprobes.make_dpt_weights(fdim, odim, hidden=16, k=k, seed=17)
    sd = {n: t.requires_grad_(True) for n, t in sd.items()}
    if kind == "depth":
        y = oprobes.depth_head(sd, feats, head, k, pt)
    else:
        y = oprobes.snorm_head(sd, feats, head, k)
    assert rel_l2(y.detach().numpy(), g[f"{name}__out"]) < TOL
    (y * T(g[f"{name}__gy"])).sum().backward()
    for n, t in sd.items():
        ref = g[f"{name}__grad__{n}"]
        assert rel_l2(t.grad.numpy(), ref) < 2e-4, n


@pytest.mark.parametrize("name,kind,src,odim", [("depth_ms_bindepth", "depth", "vit", 256), ("depth_ms_sigdepth_res", "depth", "res", 1), ("snorm_ms_ua", "snorm", "vit", 4)])
def test_multiscale_head_fwd_bwd(name, kind, src, odim):
    """MultiscaleHead (probes.py:435-458) vs outputs + grads of the reference module (golden probes_multiscale.npz)."""
    g = load_golden("probes_multiscale.npz")
    feats = [T(g["vit_feats"][i]) for i in range(4)] if src == "vit" else [T(g[f"res_feat{i}"]) for i in range(4)]
    fdim = [24] * 4 if src == "vit" else [(8, 0), (12, 0), (16, 0), (20, 0)]
    sd = {n: t.requires_grad_(True) for n, t in oprobes.make_multiscale_weights(fdim, odim, hidden=16, k=1, seed=19).items()}
    y = oprobes.depth_head(sd, feats, "multiscale", 1, "bindepth" if odim == 256 else "sigdepth") if kind == "depth" else oprobes.snorm_head(sd, feats, "multiscale", 1)
    assert str(g[f"{name}__name"]) == {"depth_ms_bindepth": "bindepth_multiscale_k1", "depth_ms_sigdepth_res": "sigdepth_multiscale_k1", "snorm_ms_ua": "snorm_multiscale_k1_UA"}[name]
    assert rel_l2(y.detach().numpy(), g[f"{name}__out"]) < TOL
    (y * T(g[f"{name}__gy"])).sum().backward()
    for n, t in sd.items():
        assert rel_l2(t.grad.numpy(), g[f"{name}__grad__{n}"]) < 2e-4, n


def test_multiscale_head_k3_fwd_bwd():
    """MultiscaleHead(kernel_size=3): un-padded convs (probes.py:400-412), every conv shrinks its map by 2 — 6x7 maps end as a 4x12
    depth map — vs outputs + grads of the reference module."""
    g = load_golden("probes_multiscale.npz")
    feats = [T(g["k3_feats"][i]) for i in range(4)]
    sd = {n: t.requires_grad_(True) for n, t in oprobes.make_multiscale_weights([24] * 4, 1, hidden=16, k=3, seed=23).items()}
    y = oprobes.depth_head(sd, feats, "multiscale", 3, "sigdepth")
    assert str(g["depth_ms_k3__name"]) == "sigdepth_multiscale_k3" and tuple(y.shape) == (2, 1, 4, 12)
    assert rel_l2(y.detach().numpy(), g["depth_ms_k3__out"]) < TOL
    (y * T(g["depth_ms_k3__gy"])).sum().backward()
    for n, t in sd.items():
        assert rel_l2(t.grad.numpy(), g[f"depth_ms_k3__grad__{n}"]) < 2e-4, n


# ------------------------------------------------------------------ losses
@pytest.mark.parametrize("B", [1, 2, 3, 5, 8, 16])
def test_depth_loss(B):
    g = load_golden("losses.npz")
    pred = T(g[f"depth_B{B}_pred"]).requires_grad_(True)
    tgt = T(g[f"depth_B{B}_target"]).clone()
    loss = olosses.depth_loss(pred, tgt)
    np.testing.assert_array_equal(tgt.numpy(), g[f"depth_B{B}_target_after"])  # quirk Q2
    assert abs(loss.item() - g[f"depth_B{B}_loss"]) < 1e-5 * abs(g[f"depth_B{B}_loss"])
    loss.backward()
    assert rel_l2(pred.grad.numpy(), g[f"depth_B{B}_grad"]) < 1e-5
    with torch.no_grad():
        assert abs(olosses.sig_loss(pred, tgt).item() - g[f"depth_B{B}_sig"]) < 1e-5
        gl = float(olosses.gradient_loss(pred, tgt))
        assert abs(gl - g[f"depth_B{B}_gradloss"]) <= 1e-5 * max(1.0, abs(g[f"depth_B{B}_gradloss"]))
    if B <= 2:
        assert g[f"depth_B{B}_gradloss"] == 0  # quirk Q1: no b/b+2 pair exists


@pytest.mark.parametrize("ua", [0, 1])
def test_angular_loss(ua):
    g = load_golden("losses.npz")
    tag = f"ang_ua{ua}"
    pred = T(g[f"{tag}_pred"]).requires_grad_(True)
    loss = olosses.angular_loss(pred, T(g[f"{tag}_gt"]), T(g[f"{tag}_mask"]), bool(ua))
    assert abs(loss.item() - g[f"{tag}_loss"]) < 1e-5
    loss.backward()
    assert rel_l2(pred.grad.numpy(), g[f"{tag}_grad"]) < 1e-5


# ------------------------------------------------------------------ schedule / optimiser
def test_schedule_table():
    g = load_golden("optim.npz")
    v = [ooptim.cosine_decay_linear_warmup(int(s), 100, 15) for s in g["sched_steps"]]
    np.testing.assert_allclose(v, g["sched_vals"], rtol=1e-12, atol=0)
    v = [ooptim.cosine_decay_linear_warmup(s, 70, 10.5) for s in range(70)]
    np.testing.assert_allclose(v, g["sched_vals_frac"], rtol=1e-12, atol=0)


def test_adamw_trajectory():
    g = load_golden("optim.npz")
    C = 16
    sd = oprobes.make_linear_head_weights([C] * 4, 256, 1, seed=5)
    tr = otrain.DepthProbeTrainer({"cls_token": torch.zeros(1, 1, C)}, sd, layers=(0, 1, 2, 3), max_step=40, warmup_step=3)
    losses = []
    for s in range(5):
        feats = [T(g[f"traj_feats{s}"][i]) for i in range(4)]
        tgt = T(g[f"traj_target{s}"]).clone()
        for p in tr.probe_sd.values():
            p.grad = None
        loss, _ = tr.forward_loss(feats, tgt)
        loss.backward()
        lr = tr.lr_at(tr.t)
        assert abs(lr - g["traj_lrs"][s]) < 1e-12
        tr.t += 1
        with torch.no_grad():
            ooptim.adamw_step([tr.probe_sd[n] for n in tr.names], [tr.probe_sd[n].grad for n in tr.names], tr.m, tr.v, tr.t, lr)
        losses.append(loss.item())
    np.testing.assert_allclose(losses, g["traj_losses"], rtol=2e-5)
    assert rel_l2(tr.probe_sd["head.conv.weight"].detach().numpy(), g["traj_final_weight"]) < 1e-5
    assert rel_l2(tr.probe_sd["head.conv.bias"].detach().numpy(), g["traj_final_bias"]) < 1e-5


def test_full_step_tiny():
    """Oracle trainer == the reference loop body (tiny ViT, 3 steps)."""
    g = load_golden("step_tiny.npz")
    D = 128
    vsd = ovit.make_vit_weights(embed_dim=D, depth=4, seed=31)
    psd = oprobes.make_linear_head_weights([D] * 4, 256, 1, seed=32)
    tr = otrain.DepthProbeTrainer(vsd, psd, layers=(0, 1, 2, 3), heads=2, max_step=30, warmup_step=2)
    losses = []
    for s in range(3):
        images, tgt = otrain.synthetic_depth_batch(4, 64, 80, rank=0, step=s)
        if s == 0:
            feats = tr.features(images)
            tr.bn_running = [(torch.zeros(D), torch.ones(D)) for _ in range(4)]
            loss, pred = tr.forward_loss(feats, tgt.clone())
            loss.backward()
            assert rel_l2(pred.detach().numpy(), g["pred0"]) < TOL
            assert rel_l2(tr.probe_sd["head.conv.weight"].grad.numpy(), g["grad_w0"]) < 1e-4
            assert rel_l2(tr.probe_sd["head.conv.bias"].grad.numpy(), g["grad_b0"]) < 1e-4
        losses.append(tr.step(images, tgt))
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-5)
    assert rel_l2(tr.probe_sd["head.conv.weight"].detach().numpy(), g["final_weight"]) < 1e-5
