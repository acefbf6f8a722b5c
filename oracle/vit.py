"""Oracle: ViT-B/16 dense multi-layer feature extraction (DINO / iBOT flavour), fp32 CPU.

Functional restatement over a flat state dict whose keys follow the DINO / iBOT
VisionTransformer (``cls_token, pos_embed, patch_embed.proj.{weight,bias},
blocks.{i}.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}.{weight,bias}``).

Test infrastructure only (see oracle/__init__.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

StateDict = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------- weights
def make_vit_weights(
    embed_dim: int = 768,
    depth: int = 12,
    mlp_ratio: float = 4.0,
    patch: int = 16,
    img: int = 224,
    seed: int = 0,
    std: float = 0.02,
    randomize_norm_bias: bool = True,
) -> StateDict:
    """Seeded random-init weights with the reference's init statistics
    (trunc-normal(0.02) linears / pos-embed / cls, ibot_transformers.py:293-309) but with
    non-trivial biases and LayerNorm affine so that every term of every kernel is
    exercised by the parity tests (a zero bias hides a missing-bias bug)."""
    g = torch.Generator().manual_seed(seed)

    def tn(*shape, s=std):
        t = torch.empty(*shape)
        torch.nn.init.trunc_normal_(t, std=s, a=-2.0, b=2.0, generator=g)
        return t

    def small(*shape, s=0.02):
        return torch.randn(*shape, generator=g) * s

    hid = int(embed_dim * mlp_ratio)
    npatch = (img // patch) ** 2
    sd: StateDict = {}
    sd["cls_token"] = tn(1, 1, embed_dim)
    sd["pos_embed"] = tn(1, npatch + 1, embed_dim)
    fan_in = 3 * patch * patch
    bound = 1.0 / math.sqrt(fan_in)
    sd["patch_embed.proj.weight"] = (torch.rand(embed_dim, 3, patch, patch, generator=g) * 2 - 1) * bound
    sd["patch_embed.proj.bias"] = (torch.rand(embed_dim, generator=g) * 2 - 1) * bound
    for i in range(depth):
        p = f"blocks.{i}."
        sd[p + "norm1.weight"] = 1.0 + (small(embed_dim, s=0.1) if randomize_norm_bias else 0)
        sd[p + "norm1.bias"] = small(embed_dim, s=0.05) if randomize_norm_bias else torch.zeros(embed_dim)
        sd[p + "attn.qkv.weight"] = tn(3 * embed_dim, embed_dim)
        sd[p + "attn.qkv.bias"] = small(3 * embed_dim) if randomize_norm_bias else torch.zeros(3 * embed_dim)
        sd[p + "attn.proj.weight"] = tn(embed_dim, embed_dim)
        sd[p + "attn.proj.bias"] = small(embed_dim) if randomize_norm_bias else torch.zeros(embed_dim)
        sd[p + "norm2.weight"] = 1.0 + (small(embed_dim, s=0.1) if randomize_norm_bias else 0)
        sd[p + "norm2.bias"] = small(embed_dim, s=0.05) if randomize_norm_bias else torch.zeros(embed_dim)
        sd[p + "mlp.fc1.weight"] = tn(hid, embed_dim)
        sd[p + "mlp.fc1.bias"] = small(hid) if randomize_norm_bias else torch.zeros(hid)
        sd[p + "mlp.fc2.weight"] = tn(embed_dim, hid)
        sd[p + "mlp.fc2.bias"] = small(embed_dim) if randomize_norm_bias else torch.zeros(embed_dim)
    sd["norm.weight"] = torch.ones(embed_dim)
    sd["norm.bias"] = torch.zeros(embed_dim)
    return sd


# --------------------------------------------------------------------------- pieces
def center_padding(images: torch.Tensor, patch: int) -> torch.Tensor:
    """evals/models/utils.py:55-72 — zero-pad H,W up to a multiple of ``patch``; the
    smaller half of the padding goes top/left."""
    h, w = images.shape[-2:]
    rh, rw = h % patch, w % patch
    if rh == 0 and rw == 0:
        return images
    # NB (reference quirk): when only one of the two dims is ragged the other still
    # receives a full ``patch`` of padding (pad = patch - 0).
    ph, pw = patch - rh, patch - rw
    top, left = ph // 2, pw // 2
    return F.pad(images, (left, pw - left, top, ph - top))


def interpolate_pos_encoding(pos_embed: torch.Tensor, npatch: int, w: int, h: int, patch: int) -> torch.Tensor:
    """ibot_transformers.py:311-336.  ``w``/``h`` are the reference's (swapped) names for
    image dims 2 and 3; bicubic with the +0.1 scale-factor nudge."""
    N = pos_embed.shape[1] - 1
    if npatch == N and w == h:
        return pos_embed
    dim = pos_embed.shape[-1]
    cls_pos = pos_embed[:, 0]
    grid = pos_embed[:, 1:]
    side = int(math.sqrt(N))
    w0 = w // patch + 0.1
    h0 = h // patch + 0.1
    grid = F.interpolate(
        grid.reshape(1, side, side, dim).permute(0, 3, 1, 2),
        scale_factor=(w0 / math.sqrt(N), h0 / math.sqrt(N)),
        mode="bicubic",
    )
    assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
    grid = grid.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((cls_pos.unsqueeze(0), grid), dim=1)


def prepare_tokens(sd: StateDict, images: torch.Tensor, patch: int = 16, pos_mode: str = "dino") -> torch.Tensor:
    """ibot_transformers.py:338-355 — conv16/16 patch embed, prepend CLS, add pos-embed.
    pos_mode "fixed": the stored table is added as is (timm ViT in mocov3.py:154-157; HF ViT-MAE
    embed_forward, mae.py:91-104 — there CLS gets cls_token + pos[0], which is the same sum)."""
    B, _, d2, d3 = images.shape
    x = F.conv2d(images, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=patch)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat((sd["cls_token"].expand(B, -1, -1), x), dim=1)
    if pos_mode == "fixed":
        return x + sd["pos_embed"]
    return x + interpolate_pos_encoding(sd["pos_embed"], x.shape[1] - 1, d2, d3, patch)


def sincos_pos_embed_2d(embed_dim: int, grid_hw, add_cls_token: bool = True) -> torch.Tensor:
    """evals/models/utils.py:75-102 (+ the HF helper it calls): 2-D sin/cos table, the first
    half of the channels encodes the w coordinate, the second half h; each half = [sin | cos] of
    pos / 10000^(2i/(D/2)); a zero row is prepended for CLS."""
    import numpy as np

    gh, gw = grid_hw
    ww, hh = np.meshgrid(np.arange(gw, dtype=np.float32), np.arange(gh, dtype=np.float32))

    def enc(pos, d):
        omega = 1.0 / 10000 ** (np.arange(d // 2, dtype=float) / (d / 2.0))
        ang = pos.reshape(-1)[:, None] * omega[None, :]
        return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)

    emb = np.concatenate([enc(ww, embed_dim // 2), enc(hh, embed_dim // 2)], axis=1)
    if add_cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return torch.from_numpy(emb).float().unsqueeze(0)


def attention(sd: StateDict, prefix: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """ibot_transformers.py:129-145 — fused QKV linear, softmax(q k^T * d^-0.5) v, proj."""
    B, N, C = x.shape
    d = C // heads
    qkv = F.linear(x, sd[prefix + "qkv.weight"], sd.get(prefix + "qkv.bias"))
    qkv = qkv.reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = (q @ k.transpose(-2, -1)) * (d ** -0.5)
    a = a.softmax(dim=-1)
    y = (a @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(y, sd[prefix + "proj.weight"], sd[prefix + "proj.bias"])


def block(sd: StateDict, i: int, x: torch.Tensor, heads: int, eps: float = 1e-6) -> torch.Tensor:
    """ibot_transformers.py:193-203 (gamma_1 is None path): pre-LN attention + MLP(GELU erf)."""
    p = f"blocks.{i}."
    C = x.shape[-1]
    y = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    x = x + attention(sd, p + "attn.", y, heads)
    y = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
    y = F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    y = F.gelu(y)
    y = F.linear(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + y


def batchnorm_tokens_train(
    x: torch.Tensor,
    weight: Optional[torch.Tensor] = None,
    bias: Optional[torch.Tensor] = None,
    running: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
    momentum: float = 0.1,
    eps: float = 1e-5,
    training: bool = True,
) -> torch.Tensor:
    """dino.py:185-191 — ``BatchNorm1d(C)`` applied to ``x.permute(0,2,1)`` ([B,C,N]) in
    train mode: per-channel mean / biased variance over all B*N tokens, CLS included;
    running stats updated with the unbiased variance (torch BatchNorm semantics)."""
    B, N, C = x.shape
    if training:
        flat = x.reshape(B * N, C)
        mean = flat.mean(dim=0)
        var = flat.var(dim=0, unbiased=False)
        if running is not None:
            n = B * N
            running[0].mul_(1 - momentum).add_(momentum * mean)
            running[1].mul_(1 - momentum).add_(momentum * var * (n / max(n - 1, 1)))
    else:
        assert running is not None
        mean, var = running
    y = (x - mean) * torch.rsqrt(var + eps)
    if weight is not None:
        y = y * weight + bias
    return y


def tokens_to_output(output: str, dense: torch.Tensor, cls: Optional[torch.Tensor], hw: Tuple[int, int]) -> torch.Tensor:
    """evals/models/utils.py:105-124."""
    h, w = hw
    if output == "cls":
        return cls
    if output == "gap":
        return dense.mean(dim=1)
    B, _, C = dense.shape
    grid = dense.reshape(B, h, w, C).permute(0, 3, 1, 2)
    if output == "dense":
        return grid.contiguous()
    if output == "dense-cls":
        return torch.cat((grid, cls[:, :, None, None].expand(-1, -1, h, w)), dim=1).contiguous()
    raise ValueError(output)


def multilayer_indices(depth: int) -> List[int]:
    """dino.py:51-57."""
    return [depth // 4 - 1, depth // 2 - 1, depth // 4 * 3 - 1, depth - 1]


# --------------------------------------------------------------------------- forward
def vit_dense_features(
    sd: StateDict,
    images: torch.Tensor,
    layers: Sequence[int],
    heads: int = 12,
    patch: int = 16,
    add_norm: bool = True,
    output: str = "dense",
    bn_affine: Optional[Sequence[Tuple[torch.Tensor, torch.Tensor]]] = None,
    bn_running: Optional[Sequence[Tuple[torch.Tensor, torch.Tensor]]] = None,
    bn_training: bool = True,
    ln_eps: float = 1e-6,
    return_tokens: bool = False,
    pos_mode: str = "dino",
    tap_input_of_block: bool = False,
    resize_to=None,
):
    """DINO.forward, dino.py:164-210 (``return_kqv`` False): pad -> tokens -> blocks with
    taps after the listed block indices (tap = optional train-mode BatchNorm1d over tokens)
    -> drop CLS -> NCHW.  The final ``vit.norm`` is never applied."""
    if resize_to is not None:  # MoCoV3.forward, mocov3.py:150-152
        images = F.interpolate(images, size=tuple(resize_to), mode="bilinear", align_corners=False)
    images = center_padding(images, patch)
    h, w = images.shape[-2] // patch, images.shape[-1] // patch
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    x = prepare_tokens(sd, images, patch, pos_mode)
    layers = list(layers)
    taps = []

    def tap_now(j, x):
        if add_norm:
            wgt, b = bn_affine[j] if bn_affine is not None else (None, None)
            run = bn_running[j] if bn_running is not None else None
            taps.append(batchnorm_tokens_train(x, wgt, b, run, training=bn_training))
        else:
            taps.append(x)

    for i in range(depth):
        if tap_input_of_block:
            # MAE.forward taps HF ``hidden_states[i]`` = the INPUT of encoder layer i (mae.py:216-217, quirk Q4)
            if i in layers:
                tap_now(layers.index(i), x)
                if len(taps) == len(layers):
                    break
            x = block(sd, i, x, heads, ln_eps)
            continue
        x = block(sd, i, x, heads, ln_eps)
        if i in layers:
            j = layers.index(i)
            if add_norm:
                wgt, b = bn_affine[j] if bn_affine is not None else (None, None)
                run = bn_running[j] if bn_running is not None else None
                taps.append(batchnorm_tokens_train(x, wgt, b, run, training=bn_training))
            else:
                taps.append(x)
            if len(taps) == len(layers):
                break
    if return_tokens:
        return taps
    n_sp = h * w
    outs = [tokens_to_output(output, t[:, -n_sp:], t[:, 0], (h, w)) for t in taps]
    return outs[0] if len(outs) == 1 else outs


def last_block_qkv(sd: StateDict, images: torch.Tensor, heads: int = 12, patch: int = 16, eps: float = 1e-6, pos_mode: str = "dino"):
    """DINO.extract_kqv, dino.py:82-131 — what the forward hook on ``blocks[-1].attn.qkv`` captures: the fused qkv projection of the
    last block's pre-normalised input, after all earlier blocks (no centre padding on this path: dino.py:105 calls prepare_tokens
    directly).  Returns (q, k, v) as [B, N, C] each (heads merged back: dino.py:117-124)."""
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    x = prepare_tokens(sd, images, patch, pos_mode)
    for i in range(depth - 1):
        x = block(sd, i, x, heads, eps)
    p = f"blocks.{depth - 1}."
    C = x.shape[-1]
    y = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    qkv = F.linear(y, sd[p + "attn.qkv.weight"], sd.get(p + "attn.qkv.bias"))
    B, N, _ = qkv.shape
    qkv = qkv.reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    return tuple(t.transpose(1, 2).reshape(B, N, C) for t in (qkv[0], qkv[1], qkv[2]))


def dino_kqv_features(sd: StateDict, images: torch.Tensor, fixed_size: int, mode: str = "k", heads: int = 12, patch: int = 16) -> torch.Tensor:
    """DINO.forward with return_kqv=True, dino.py:82-168: torchvision ``Resize((fixed, fixed))`` (bilinear, antialias: the default for
    tensors in the pinned torchvision 0.17.1) -> extract_kqv -> the selected projection(s) without the CLS row as [B, C (3C), h*w]."""
    if images.ndim == 3:
        images = images[None]
    x = F.interpolate(images.float(), size=(fixed_size, fixed_size), mode="bilinear", align_corners=False, antialias=True)
    q, k, v = last_block_qkv(sd, x, heads, patch)
    B, _, C = k.shape
    hw = (fixed_size // patch) ** 2
    pick = {"k": [k], "q": [q], "v": [v], "kqv": [k, q, v]}[mode]
    return torch.cat([t[:, 1:].transpose(1, 2).reshape(B, C, hw) for t in pick], dim=1)
