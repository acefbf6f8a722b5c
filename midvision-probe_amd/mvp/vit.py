"""ViT-B/16-class frozen backbone forward on the HIP kernels (DINO / iBOT / MoCo-v3 / MAE
share this engine; the wrappers under evals/models differ in weights, pos-embed policy,
LayerNorm eps and which block outputs are tapped).

Data layout in HBM (per batch of B images, N = 1 + gh*gw tokens, M = B*N rows):
  x        fp32  [M, C]      residual stream (kept fp32: LN statistics and residual adds)
  xn       bf16 pair [M, C]  LayerNorm output = A operand of the next GEMM
  qkv      bf16 pair [M, 3C] fused projection, read in place by the attention kernel
  ao       bf16 pair [M, C]  attention output (already in (B, N, H*64) order)
  hmid     bf16 pair [M, 4C] fc1+GELU output
  weights  bf16 pairs, torch Linear layout [N_out, K] (K contiguous) — split once at load.
Everything is allocated once per (B, gh, gw) and reused: the frozen forward allocates only
the returned NCHW maps.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import lib, ops, pipeline
from .lib import PREC_BF16, PREC_BF16X3


def parse_precision(p) -> int:
    """'bf16x3' (default: three bf16 products per contraction, ~1e-5 on the features), 'f16x2' (opt-in: the four GEMMs of every block
    with two fp16 products over compensated fp16 pairs, include/mvp_hip.h MVP_PREC_F16X2; everything else as in bf16x3; the same
    ~1e-5 on the features, |activation| <= 65504), 'bf16' (one product: fails the 1e-3 feature contract)."""
    if p in (PREC_BF16, PREC_BF16X3, lib.PREC_F16X2):
        return p
    s = str(p).lower()
    if s in ("bf16", "fast"):
        return PREC_BF16
    if s in ("bf16x3", "x3", "exact", "fp32"):
        return PREC_BF16X3
    if s in ("f16x2", "fp16x2", "x2"):
        return lib.PREC_F16X2
    raise ValueError(f"unknown precision {p!r} (use 'bf16x3', 'f16x2' or 'bf16')")


class TapOutputs(list):
    """The list of NCHW maps a backbone returns, plus (privately) the token-major bf16
    packing of the same features that the probe-head GEMMs consume."""
    packed = None


class TapGroups(list):
    """What a GROUPED forward returns (``forward_taps(..., groups=G)``): one ``TapOutputs`` per batch of the group, in batch order."""


class PackedFeatures:
    """Token-major operand of the linear-probe GEMMs: F [Mpad, Cpad] bf16 pair (forward: F·Wᵀ by the NT GEMM;
    weight gradient: gᵀ·F by the TN split-K kernel, which reads the same row-major image through transposed LDS reads)."""

    def __init__(self, B, h, w, Ctot, precision, device, tok=None):
        self.B, self.h, self.w, self.Ctot, self.precision = B, h, w, Ctot, precision
        self.M = B * h * w
        self.Mpad, self.Cpad = self.padded(B, h, w, Ctot)
        # ``tok``: a zero-initialised [Mpad, Cpad] pair owned by the caller (the batches of a grouped forward share one allocation)
        self.tok = tok if tok is not None else ops.zeros_pair((self.Mpad, self.Cpad), precision, device)
        self.sources: List[Tuple[int, int]] = []  # (data_ptr, _version) of the NCHW maps packed here
        self.source_refs: List[torch.Tensor] = []  # the maps themselves: while they live, their addresses cannot be recycled
        self.generation = 0  # bumped every time the buffers are rewritten (they are reused across steps)
        self.scratch: Dict[str, object] = {}  # per-shape scratch of the head backward (zero-padded once)


def _padded(B, h, w, Ctot):
    """(Mpad, Cpad) of a packing: NT GEMM K % 64, TN kernel Cin % 128; pad rows / columns stay zero."""
    return (B * h * w + 63) // 64 * 64, (Ctot + 127) // 128 * 128


PackedFeatures.padded = staticmethod(_padded)

_PACK_REGISTRY: Dict[int, PackedFeatures] = {}
# one entry per (pipeline slot, forward shape, batch of the forward): a span pipeline keeps 2 slots x two shapes (floor / ceil of T / B
# whole batches) x up to 14 batches = 54 entries at B = 8, and a model has a train and an eval pipeline; a graph replay re-registers
# its packings (pipeline._forward), so an entry that ages out anyway comes back at its forward's next replay
ATT_QK_DEFAULT = "auto"  # MVP_ATT_QK: "f16" = Q.K^T in two f16 products (ViTEngine.att_qk_f16), "pair" = three bf16 products, "auto" = f16 for f16x2 engines
_PACK_REGISTRY_MAX = 256


def _ver(t: torch.Tensor) -> int:
    try:
        return t._version
    except RuntimeError:  # inference tensors do not track a version counter
        return -1


def register_pack(maps: Sequence[torch.Tensor], pack: PackedFeatures) -> None:
    """The cache key is (data_ptr, _version) of each map.  The registry also KEEPS the maps (strong references, dropped at the next
    backbone forward): were they freed, the caching allocator could hand the same addresses to unrelated same-shape tensors with
    version 0 (re-uploaded cached features, clones), which would then silently alias this packing."""
    pack.sources = [(m.data_ptr(), _ver(m)) for m in maps]
    pack.source_refs = list(maps)
    # one entry per packing buffer: the probe consumes features right after the backbone, or — with several forwards in flight
    # (mvp/pipeline.py) — one entry per pipeline slot.  Entries of other engines / shapes age out beyond the newest few.
    for k in [k for k, v in _PACK_REGISTRY.items() if v is pack]:
        del _PACK_REGISTRY[k]
    _PACK_REGISTRY[maps[0].data_ptr()] = pack
    while len(_PACK_REGISTRY) > _PACK_REGISTRY_MAX:
        del _PACK_REGISTRY[next(iter(_PACK_REGISTRY))]


def lookup_pack(maps: Sequence[torch.Tensor]) -> Optional[PackedFeatures]:
    """Return the packing produced alongside ``maps`` if these are still the very same
    (unmodified) buffers; ``.detach()`` in the training loop keeps data_ptr and version."""
    if not maps:
        return None
    pack = _PACK_REGISTRY.get(maps[0].data_ptr())
    if pack is None or len(pack.sources) != len(maps):
        return None
    for m, (p, v), src in zip(maps, pack.sources, pack.source_refs):
        if m.data_ptr() != p or _ver(m) != v or m.shape != src.shape or m.dtype != src.dtype or m.stride() != src.stride():
            return None
    return pack


class ViTEngine:
    def __init__(self, state_dict: Dict[str, torch.Tensor], *, heads: int, patch: int = 16, ln_eps: float = 1e-6,
                 precision="bf16x3", device="cuda", pos_embed_mode: str = "dino", qkv_fused: bool = True):
        self.device = torch.device(device)
        self.precision = parse_precision(precision)
        # 'f16x2': a bf16x3 engine (buffers, patch embedding, attention, taps) whose four block GEMMs run two products (lib.PREC_F16X2)
        self.f16x2 = self.precision == lib.PREC_F16X2
        if self.f16x2:
            self.precision = PREC_BF16X3
        self.heads, self.patch, self.ln_eps = heads, patch, ln_eps
        self.att_v_f16 = self.precision == lib.PREC_BF16X3 and os.environ.get("MVP_ATT_V", "f16") != "pair"
        # Q.K^T as TWO f16 products over compensated fp16 pairs (Q: activation form, K: weight-side form, both written by the qkv GEMM's
        # epilogue; csrc/attention.hip, VF16 == 2) instead of three bf16 ones; MVP_ATT_QK=pair brings the bf16 pairs back
        # Default: on for an f16x2 engine (whose activations are bound to fp16's range anyway), off for bf16x3 (which keeps fp32's exponent
        # range for Q and K).  +0.4 % at 224^2, +1.3 % at 480x640 (profiles/r04_qk_ab.txt); the reference's goldens read the same 1.5e-5 ... 2.4e-5.
        qk = os.environ.get("MVP_ATT_QK", ATT_QK_DEFAULT)
        self.att_qk_f16 = self.att_v_f16 and (qk == "f16" or (qk == "auto" and self.f16x2))
        self.check_f16_range = os.environ.get("MVP_CHECK_F16_RANGE", "0") == "1"  # diagnostic: see _check_f16_range
        self.pos_embed_mode = pos_embed_mode
        sd = {k: v.detach().to(self.device, torch.float32).contiguous() for k, v in state_dict.items()}
        self.C = sd["cls_token"].shape[-1]
        if self.C != heads * 64:
            raise lib.MvpError(f"attention kernel requires head_dim 64 (C={self.C}, heads={heads})")
        self.depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
        self.cls = sd["cls_token"].reshape(-1).contiguous()
        self.pos_embed = sd["pos_embed"]  # [1, 1+n, C] fp32
        pw = sd["patch_embed.proj.weight"]
        self.in_chans = pw.shape[1]
        self.w_patch = ops.split_bf16(pw.reshape(self.C, -1), self.precision)
        self.b_patch = sd["patch_embed.proj.bias"]
        self.blocks = []
        for i in range(self.depth):
            p = f"blocks.{i}."
            blk = dict(
                n1w=sd[p + "norm1.weight"], n1b=sd[p + "norm1.bias"],
                qkv_w=ops.split_bf16(sd[p + "attn.qkv.weight"], self.precision),
                qkv_b=sd.get(p + "attn.qkv.bias"),
                proj_w=ops.split_bf16(sd[p + "attn.proj.weight"], self.precision), proj_b=sd[p + "attn.proj.bias"],
                n2w=sd[p + "norm2.weight"], n2b=sd[p + "norm2.bias"],
                fc1_w=ops.split_bf16(sd[p + "mlp.fc1.weight"], self.precision), fc1_b=sd[p + "mlp.fc1.bias"],
                fc2_w=ops.split_bf16(sd[p + "mlp.fc2.weight"], self.precision), fc2_b=sd[p + "mlp.fc2.bias"],
            )
            if self.f16x2:  # the weight operands of the two-product GEMMs (ops.f16x2_weight: compensated fp16 pairs)
                for n in ("qkv_w", "proj_w", "fc1_w", "fc2_w"):
                    blk[n] = ops.f16x2_weight(sd[p + {"qkv_w": "attn.qkv.weight", "proj_w": "attn.proj.weight", "fc1_w": "mlp.fc1.weight", "fc2_w": "mlp.fc2.weight"}[n]])
            if self.precision == PREC_BF16X3:  # hi|lo-interleaved copies of the frozen weights for the large-M GEMM kernel (mvp.ops.interleave_pair)
                for n in ("qkv_w", "proj_w", "fc1_w", "fc2_w"):
                    blk[n + "_ilv"] = ops.interleave_pair(blk[n])
            self.blocks.append(blk)
        self.hidden = self.blocks[0]["fc1_b"].numel()
        self._ws: Dict[Tuple[int, int, int, int], dict] = {}  # (B, gh, gw, pipeline slot)
        self._pos: Dict[Tuple[int, int], torch.Tensor] = {}
        self._packs: Dict[Tuple[int, int, int, int, int], PackedFeatures] = {}
        self._slot_outs: Dict[tuple, dict] = {}  # output maps of pipelined forwards, owned by the slot
        self._ns_lru: List[int] = []  # pipeline namespaces, least recently used first (see _touch_namespace)
        self._carry: Dict[tuple, torch.Tensor] = {}  # (stream, batch, gh, gw, taps) -> [taps, batch * N, C] tap-level rows of a batch cut by a span's end
        pipeline.publish()  # the split weights are read by forwards on any stream

    # ------------------------------------------------------------------ helpers
    def _touch_namespace(self) -> None:
        """Slot keys are (pipeline namespace, slot index): every FeaturePipeline owns its buffer sets, so two pipelines over this
        backbone (a training loop suspended with forwards in flight, a validation pass) never write each other's.  A loop that builds
        a new pipeline per epoch would otherwise pile the sets up: keep those of the two most recently used namespaces (a captured
        graph holds on to its own slot's buffers whatever happens here)."""
        slot = pipeline.current_slot()
        ns = slot[0] if isinstance(slot, tuple) else None
        if ns is None or (self._ns_lru and self._ns_lru[-1] == ns):
            return
        if ns in self._ns_lru:
            self._ns_lru.remove(ns)
        self._ns_lru.append(ns)
        if len(self._ns_lru) > 2:
            dead = set(self._ns_lru[:-2])
            self._ns_lru = self._ns_lru[-2:]
            gone = lambda k: isinstance(k[-1], tuple) and k[-1][0] in dead  # noqa: E731
            self._ws = {k: v for k, v in self._ws.items() if not gone(k)}
            self._packs = {k: v for k, v in self._packs.items() if not gone(k)}
            self._slot_outs = {k: v for k, v in self._slot_outs.items() if not gone(k)}

    def _workspace(self, B: int, gh: int, gw: int, headroom: int = 0) -> dict:
        self._touch_namespace()
        key = (B, gh, gw, pipeline.current_slot())
        ws = self._ws.get(key)
        if ws is not None and ws["headroom"] < headroom:
            ws = None
        if ws is None:
            N = 1 + gh * gw
            M = B * N
            C, dev, pr = self.C, self.device, self.precision
            # When every GEMM of a block goes to the large-M kernel (M fills whole rounds of 256x256 tiles: grouped forwards), its A
            # operands — LayerNorm output, attention output, fc1 output — are kept as hi|lo-interleaved arrays (ops.IlvPair): a 32-deep
            # k-step of a row is then one whole 128-byte line for the LDS-DMA (2-3 % per GEMM on top of the interleaved weights).
            ilv = (pr == PREC_BF16X3 and os.environ.get("MVP_ILV", "1") != "0" and C % 32 == 0 and self.hidden % 32 == 0 and
                   all(ops.gemm_tile(M, n, k, pr, 1, pipeline.tile_policy()).startswith("pp ") for n, k in ((3 * C, C), (C, C), (self.hidden, C), (C, self.hidden))))
            xfull = torch.empty(M + headroom * N, C, dtype=torch.float32, device=dev)
            ws = dict(
                xfull=xfull, headroom=headroom, x=xfull[headroom * N:],
                xn=ops.IlvPair(M, C, dev) if ilv else ops.empty_pair((M, C), pr, dev),
                qkv=ops.empty_pair((M, 3 * C), pr, dev),
                ao=ops.IlvPair(M, C, dev) if ilv else ops.empty_pair((M, C), pr, dev),
                hmid=ops.IlvPair(M, self.hidden, dev) if ilv else ops.empty_pair((M, self.hidden), pr, dev),
                patches=ops.empty_pair((B * gh * gw, self.in_chans * self.patch * self.patch), pr, dev),
            )
            self._ws = {k: v for k, v in self._ws.items() if k[:3] == key[:3]}  # keep one resolution resident (one buffer set per slot)
            self._ws[key] = ws
        return ws

    def slot_state(self, slot: int) -> list:
        """Every buffer set this engine currently keeps for pipeline slot ``slot`` (activation workspaces, feature packings, output
        maps).  A captured hipGraph of that slot's forward holds their raw addresses: the pipeline keeps this list alive with the
        graph, because the engine itself drops the buffers of other resolutions when a new one arrives."""
        return ([v for k, v in self._ws.items() if k[-1] == slot] + [v for k, v in self._packs.items() if k[-1] == slot]
                + [v for k, v in self._slot_outs.items() if k[-1] == slot] + list(self._pos.values()) + list(self._carry.values()))

    def pos_for(self, gh: int, gw: int, dim2: int, dim3: int) -> torch.Tensor:
        """Pos-embed for a gh x gw grid.  'dino': bicubic resize with the +0.1 scale nudge of
        ibot_transformers.py:311-336 (done once per resolution, cached; torch's bicubic on the
        device is used for this one-time [1,C,14,14] resample)."""
        key = (gh, gw)
        pe = self._pos.get(key)
        if pe is not None:
            return pe
        n = self.pos_embed.shape[1] - 1
        if self.pos_embed_mode == "fixed" or (gh * gw == n and dim2 == dim3):
            pe = self.pos_embed[0].contiguous()
        else:
            side = int(math.sqrt(n))
            w0, h0 = dim2 // self.patch + 0.1, dim3 // self.patch + 0.1
            grid = self.pos_embed[:, 1:].reshape(1, side, side, self.C).permute(0, 3, 1, 2)
            grid = F.interpolate(grid, scale_factor=(w0 / math.sqrt(n), h0 / math.sqrt(n)), mode="bicubic")
            assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
            pe = torch.cat((self.pos_embed[0, :1], grid.permute(0, 2, 3, 1).reshape(-1, self.C)), dim=0).contiguous()
        self._pos[key] = pe
        pipeline.publish()
        return pe

    def set_pos_embed(self, pos_embed: torch.Tensor) -> None:
        self.pos_embed = pos_embed.detach().to(self.device, torch.float32).contiguous()
        self._pos.clear()

    # ------------------------------------------------------------------ forward
    def tokens(self, images: torch.Tensor, headroom: int = 0) -> Tuple[dict, int, int, int]:
        """Patch-embed + CLS + pos-embed into ws['x'] (K1); returns (ws, B, gh, gw).  ``headroom``: images' worth of rows kept free in
        front of ws['x'] (ws['xfull'] = head-room + x; span forwards, see forward_taps)."""
        images = images.to(self.device, torch.float32).contiguous()
        B, Cin, H, W = images.shape
        P = self.patch
        rh, rw = H % P, W % P
        if rh == 0 and rw == 0:
            ph = pw = 0
        else:  # center_padding quirk: a non-ragged dim still gets a full patch (utils.py:55-72)
            ph, pw = P - rh, P - rw
        gh, gw = (H + ph) // P, (W + pw) // P
        ws = self._workspace(B, gh, gw, headroom)
        N, C = 1 + gh * gw, self.C
        ops.patch_gather(images, ws["patches"], P, gh, gw, ph // 2, pw // 2)
        pos = self.pos_for(gh, gw, H + ph, W + pw)
        Kp = Cin * P * P
        # x[b, 1+p, :] = patches · Wᵀ + bias + pos[1+p]   (row remap skips the CLS slot)
        ops.gemm(ws["patches"], self.w_patch, B * gh * gw, C, Kp, bias=self.b_patch, residual=pos[1:], out_f32=ws["x"],
                 precision=self.precision, row_group=gh * gw, row_group_stride=N, row_group_off=1, res_row_mod=gh * gw)
        ops.cls_rows(self.cls, pos, ws["x"], B, N, C)
        return ws, B, gh, gw

    def _check_f16_range(self, what: str, pair, rows: int) -> None:
        """MVP_CHECK_F16_RANGE=1 (diagnostic, synchronises): an fp16 half at +-65504 means an activation left the range of the two-product
        mode (its pairs saturate instead of overflowing, so the result would be wrong without a NaN to show for it)."""
        for t in ([pair.t[:rows]] if isinstance(pair, ops.IlvPair) else [pair[0][:rows], pair[1][:rows]]):
            if not bool((t.view(torch.float16).abs() < 65504.0).all()):  # (NaN fails the comparison too)
                raise lib.MvpError(f"f16x2: {what} holds values beyond fp16's range (|v| >= 65504): run this model with precision='bf16x3' (MVP_PRECISION=bf16x3)")

    def run_block(self, i: int, ws: dict, B: int, N: int) -> None:
        blk, C, M, pr = self.blocks[i], self.C, B * N, self.precision
        f2 = self.f16x2
        chk = f2 and self.check_f16_range
        gp = lib.PREC_F16X2 if f2 else pr  # precision of the four block GEMMs
        x = ws["x"]
        ops.layernorm(x, blk["n1w"], blk["n1b"], ws["xn"], M, C, self.ln_eps, out_f16=f2)
        if chk:
            self._check_f16_range(f"block {i}: LayerNorm 1 output", ws["xn"], M)
        # bf16x3: the V third of qkv leaves the GEMM as hi = fp16, lo = bf16, and the attention kernel holds its probabilities as one
        # fp16 value (csrc/attention.hip, VF16; MVP_ATT_V=pair brings back the bf16-pair probabilities of rounds 1-3)
        vf16 = self.att_v_f16
        # (f16x2: Q and K leave as compensated fp16 pairs too — activation / weight-side form, Q.K^T in two f16 products — unless MVP_ATT_QK=pair;
        #  the attention output, LayerNorm's and fc1's output leave as compensated fp16 activation pairs)
        qk16 = self.att_qk_f16
        ops.gemm(ws["xn"], blk["qkv_w"], M, 3 * C, C, bias=blk["qkv_b"], out=ws["qkv"], precision=gp, w_ilv=blk.get("qkv_w_ilv"),
                 f16_col0=(-2 * C if qk16 else 2 * C) if vf16 else 0)
        if chk and qk16:
            self._check_f16_range(f"block {i}: Q / K", (ws["qkv"][0][:, :2 * C], ws["qkv"][1][:, :2 * C]), M)
        ops.attention(ws["qkv"], ws["ao"], B, N, self.heads, 64 ** -0.5, pr, v_f16=vf16, qk_f16=qk16, out_f16=f2)
        if chk:
            self._check_f16_range(f"block {i}: attention output", ws["ao"], M)
        ops.gemm(ws["ao"], blk["proj_w"], M, C, C, bias=blk["proj_b"], residual=x, out_f32=x, precision=gp, w_ilv=blk.get("proj_w_ilv"))
        ops.layernorm(x, blk["n2w"], blk["n2b"], ws["xn"], M, C, self.ln_eps, out_f16=f2)
        if chk:
            self._check_f16_range(f"block {i}: LayerNorm 2 output", ws["xn"], M)
        ops.gemm(ws["xn"], blk["fc1_w"], M, self.hidden, C, bias=blk["fc1_b"], out=ws["hmid"], act=lib.ACT_GELU, precision=gp, w_ilv=blk.get("fc1_w_ilv"),
                 f16_col0=-1 if f2 else 0)
        if chk:
            self._check_f16_range(f"block {i}: GELU(fc1) output", ws["hmid"], M)
        ops.gemm(ws["hmid"], blk["fc2_w"], M, C, self.hidden, bias=blk["fc2_b"], residual=x, out_f32=x, precision=gp, w_ilv=blk.get("fc2_w_ilv"))

    def forward_taps(self, images: torch.Tensor, layers: Sequence[int], *, bn: Optional[Sequence[dict]] = None,
                     bn_mode: int = 0, pack: bool = True, tap_input_of_block: bool = False, want_cls: bool = False, groups: int = 1):
        """Run blocks up to the last tapped one; at each tap apply the (train-mode) tap BN and
        emit the NCHW map (+ token-major packing).  ``bn[j]`` = dict(weight,bias,running_mean,
        running_var) tensors or None; bn_mode: 0 train stats, 1 eval, 2 no norm.
        ``tap_input_of_block``: tap the INPUT of block i instead of its output (HF
        hidden_states indexing used by the MAE wrapper, quirk Q4).
        ``groups`` = G > 1: ``images`` holds G batches of equal size stacked along dim 0 (mvp/pipeline.py).  Patch embedding,
        LayerNorm, the GEMMs and attention are per-row / per-image, so the G batches simply share their launches (M = G * B * N
        rows: the large-M GEMM kernel); the tap BN — train-mode statistics over ONE batch (dino.py:185-191) — runs per batch on that
        batch's rows.  Every batch gets exactly the bits it would get alone; returns ``TapGroups`` (one ``TapOutputs`` per batch).
        ``groups`` = ``pipeline.Span(batch, carry)``: the forward's images are a SPAN of the image stream that need not start or end on
        a batch boundary (batches of ``batch`` images; the first ``batch - carry`` images complete the batch whose first ``carry``
        images ended the previous span, when carry > 0).  The blocks do not care; per tap, the carried images' rows (kept in
        ``self._carry``, written by the previous span's forward on the same stream) are copied in FRONT of this span's rows — the
        ``x`` workspace has that head-room — so that all complete batches are contiguous and one grouped tap-BN launch serves them, and
        the rows of a trailing incomplete batch are copied to the carry store for the next span.  Returns ``TapGroups`` of the
        (carry + images) // batch batches this forward completes."""
        span = groups if isinstance(groups, pipeline.Span) else None
        ws, Bt, gh, gw = self.tokens(images, headroom=span.batch if span else 0)
        N, C, hw = 1 + gh * gw, self.C, gh * gw
        if span is not None:
            B, carry = int(span.batch), int(span.carry)
            if B < 1 or not 0 <= carry < B:
                raise lib.MvpError(f"span forward: carry {carry} outside [0, {B})")
            G, tail = (carry + Bt) // B, (carry + Bt) % B
            if G < 1:
                raise lib.MvpError(f"span forward: {carry} + {Bt} images complete no batch of {B}")
            x_bn = ws["xfull"][(ws["headroom"] - carry) * N:]  # the complete batches: carried rows (copied per tap) + this span's rows
            ckey = (span.stream, B, gh, gw, len(list(layers)))  # one store per image stream (pipeline): see pipeline.Span
            store = self._carry.get(ckey)
            if store is None:
                for k in [k for k in self._carry if k[0] == span.stream]:
                    del self._carry[k]  # the stream changed shape: its old store has no reader left (captured graphs keep theirs alive)
                for k in [k for k in self._carry if k[0] not in self._ns_lru]:
                    del self._carry[k]  # (streams of pipelines whose buffer sets were dropped too: _touch_namespace)
                store = self._carry[ckey] = torch.empty(len(list(layers)), B * N, C, dtype=torch.float32, device=self.device)
        else:
            if groups < 1 or Bt % groups:
                raise lib.MvpError(f"grouped forward: {Bt} images do not split into {groups} equal batches")
            G, B, carry, tail = groups, Bt // groups, 0, 0
            x_bn = ws["x"]
        layers = list(layers)
        outs_g = [TapOutputs() for _ in range(G)]
        bn_ws = ws.get(("bn_ws", G))  # tap-BN partials + scale / shift of G batches of B * N rows
        if bn_ws is None:
            bn_ws = ws[("bn_ws", G)] = torch.empty(G * ops.bn_tokens_workspace_bytes(B * N, C) // 4 + 16, dtype=torch.float32, device=self.device)
        packed_g = [None] * G
        if pack:  # reuse the (zero padded) packing buffers across steps: only the valid region is rewritten
            pkey = (B, gh, gw, len(layers), G, pipeline.current_slot())
            packs = self._packs.get(pkey)
            if packs is None:
                Mpad, Cpad = PackedFeatures.padded(B, gh, gw, C * len(layers))
                big = ops.zeros_pair((G, Mpad, Cpad), self.precision, self.device)  # one allocation: the tap kernel writes all batches in one launch
                packs = [PackedFeatures(B, gh, gw, C * len(layers), self.precision, self.device,
                                        tok=(big[0][g], big[1][g] if big[1] is not None else None)) for g in range(G)]
                self._packs = {k: v for k, v in self._packs.items() if k[:4] == pkey[:4]}
                self._packs[pkey] = packs
            for g in range(G):
                packs[g].generation += 1
            packed_g = list(packs)
        # Plain calls return freshly allocated maps (the caller may keep them).  A pipelined forward (mvp/pipeline.py) writes into
        # buffers owned by its slot instead — valid until the slot's next forward, which is the pipeline's contract — so the
        # steady state allocates nothing and no block ever changes hands between the side stream's and the trainer's allocator pools.
        def new_out():
            return dict(stats=torch.empty(G, len(layers), 3 * C, dtype=torch.float32, device=self.device),
                        nchw=[torch.empty(G, B, C, gh, gw, dtype=torch.float32, device=self.device) for _ in layers],
                        cls=[torch.empty(G, B, C, dtype=torch.float32, device=self.device) if want_cls else None for _ in layers])

        if pipeline.pipelined():
            okey = (B, gh, gw, tuple(layers), bool(want_cls), G, pipeline.current_slot())
            out = self._slot_outs.get(okey)
            if out is None:
                out = new_out()
                self._slot_outs = {k: v for k, v in self._slot_outs.items() if k[:5] == okey[:5]}
                self._slot_outs[okey] = out
        else:
            out = new_out()
        for o in outs_g:
            o.cls = []

        # Train-mode tap BN updates its running statistics in place: the only state a frozen forward mutates.  Forwards in flight on
        # different streams finish in any order, so a pipelined forward leaves that update to the consumer (pipeline.defer), which
        # applies it on the trainer's stream in batch order — same arithmetic, same bits (mvp_bn_running_update).
        defer = bn is not None and bn_mode == 0 and pipeline.pipelined()
        if (G > 1 or span is not None) and bn is not None and bn_mode == 0 and not defer:
            raise lib.MvpError("a grouped forward with train-mode tap BN must run under the pipeline (its running-statistics updates are per batch)")

        running = [[] for _ in range(G)]  # per batch of the group: its taps' deferred running-statistics updates

        def tap(j):
            b = bn[j] if bn is not None else None
            nchw, cls = out["nchw"][j], out["cls"][j]
            tok0 = packed_g[0].tok if pack else None
            if carry:
                x_bn[:carry * N].copy_(store[j, :carry * N])
            if tail:
                store[j, :tail * N].copy_(ws["x"][(Bt - tail) * N:Bt * N])
            # ONE call for all batches of the group: statistics, normalisation and outputs per batch (mvp_bn_tokens_args.groups)
            ops.bn_tokens_to_nchw(
                x_bn, B, N, C, hw, workspace=bn_ws, stats=out["stats"][0, j],
                gamma=b["weight"] if b else None, beta=b["bias"] if b else None,
                running_mean=b["running_mean"] if b else None, running_var=b["running_var"] if b else None,
                nchw=nchw[0], tok=tok0, ld_tok=packed_g[0].Cpad if pack else 0, col_off=j * C,
                mode=bn_mode, cls_out=cls[0] if want_cls else None, num_batches_tracked=b.get("num_batches_tracked") if b else None, defer_running=defer,
                groups=G, stats_gstride=out["stats"].stride(0), nchw_gstride=nchw.stride(0),
                tok_gstride=(packed_g[0].Mpad * packed_g[0].Cpad) if pack else 0, cls_gstride=cls.stride(0) if want_cls else 0)
            for g in range(G):
                if want_cls:
                    outs_g[g].cls.append(cls[g])
                if defer and b is not None and b.get("running_mean") is not None:
                    running[g].append((out["stats"][g, j], b["running_mean"], b["running_var"], b.get("num_batches_tracked"), C))
                outs_g[g].append(nchw[g])

        for i in range(self.depth):
            if tap_input_of_block and i in layers:
                tap(layers.index(i))
                if len(outs_g[0]) == len(layers):
                    break
            self.run_block(i, ws, Bt, N)
            if (not tap_input_of_block) and i in layers:
                tap(layers.index(i))
                if len(outs_g[0]) == len(layers):
                    break
        for g in range(G):
            if running[g]:  # all taps of a batch in ONE launch on the consumer's stream (mvp_bn_running_update_n; the modules are distinct)
                pipeline.defer(lambda items=running[g]: ops.bn_running_update_many(items), group=g)
            outs_g[g].stats = out["stats"][g]
            if packed_g[g] is not None:
                outs_g[g].packed = packed_g[g]
                register_pack(outs_g[g], packed_g[g])
        return outs_g[0] if (G == 1 and span is None) else TapGroups(outs_g)

    def last_block_qkv(self, images: torch.Tensor) -> torch.Tensor:
        """The fused qkv projection of the LAST block (fp32 [B, N, 3C]: q | k | v, heads side by side) — what the reference captures
        with a forward hook on ``blocks[-1].attn.qkv`` (dino.py:82-113).  Blocks 0 .. depth-2 run as usual; the last block stops after
        LayerNorm 1 and the projection (its attention output is never used on that path)."""
        ws, B, gh, gw = self.tokens(images)
        N = 1 + gh * gw
        for i in range(self.depth - 1):
            self.run_block(i, ws, B, N)
        blk, C, M = self.blocks[self.depth - 1], self.C, B * N
        f2 = self.f16x2
        ops.layernorm(ws["x"], blk["n1w"], blk["n1b"], ws["xn"], M, C, self.ln_eps, out_f16=f2)
        out = torch.empty(M, 3 * C, dtype=torch.float32, device=self.device)
        ops.gemm(ws["xn"], blk["qkv_w"], M, 3 * C, C, bias=blk["qkv_b"], out_f32=out, precision=lib.PREC_F16X2 if f2 else self.precision,
                 w_ilv=blk.get("qkv_w_ilv"))
        return out.view(B, N, 3 * C)

    def forward_tokens(self, images: torch.Tensor, n_blocks: Optional[int] = None) -> torch.Tensor:
        """Raw fp32 token stream after ``n_blocks`` blocks ([B, N, C]); for tests / CLS outputs."""
        ws, B, gh, gw = self.tokens(images)
        N = 1 + gh * gw
        for i in range(self.depth if n_blocks is None else n_blocks):
            self.run_block(i, ws, B, N)
        return ws["x"].view(B, N, self.C).clone()
