"""Frozen-backbone forwards in flight on side HIP streams.

The backbone is frozen (train_depth.py:103-112 runs it under no_grad and detaches), so the features of batch t+1 do not
depend on probe step t.  ``FeaturePipeline`` launches each forward on one of ``depth`` side streams as soon as its batch is
known; the trainer's stream picks the features up with an event wait.  Two independent kernel chains then share the chip: the
GEMM tiles of one fill the CUs the other leaves idle (600 tiles on 256 CUs), and a chain's LayerNorm / attention / probe
kernels run under the other's GEMMs.  Measured on MI355X (tools/micro/two_stream_probe.py, ViT-B/16 4-tap forward, B=16
each): two forwards back to back 5.34 ms, the same two on two streams 4.61 ms.

Nothing about the arithmetic changes: every batch still runs alone through the same kernels with its own tap-BN batch
statistics, and the tap-BN running statistics are updated in batch order (ViTEngine orders tap j of forward t+1 after tap j
of forward t with an event).  A pipelined run is bit-identical to the serial one (tests/test_gpu_pipeline.py).

Each in-flight forward owns a *slot*: the engine keeps one activation workspace, one token-major feature packing and one set
of output maps per slot, so a forward never writes buffers the probe step of an earlier batch is still reading, and the steady
state allocates nothing.  A slot is reused only after the trainer's stream has passed the probe step that consumed it
(``submit`` makes the side stream wait for the trainer's stream).  Contract: the features ``next()`` returns are valid until
``depth`` further forwards have been submitted — consume them (probe forward + backward) before feeding the pipeline again, as
``pipelined_features`` does; ``.clone()`` anything that must live longer.

Grouped forwards (round 3).  A forward may cover a GROUP of G consecutive batches: their images are stacked into one
[G * B, 3, H, W] input and run through ONE chain of launches with M = G * B * N token rows — what the large-M GEMM kernel
(csrc/gemm_pp.hip: 256x256 tiles, 92 % of the MFMA issue rate in its main loop) needs to fill 256 CUs: at B = 16, 224^2 a single
batch has 3152 rows = 13 x 3 such tiles, six batches have 74 x 3.  Every kernel of the frozen forward is per row or per image
except the tap BN, whose train-mode statistics belong to ONE batch (dino.py:185-191): the engine runs it per batch on that
batch's rows, and its running-statistics updates are applied per batch, in batch order, when the batch is handed over.  Each batch
therefore gets exactly the bits it would get alone (tests/test_gpu_pipeline.py); the probe steps still run one batch at a time, in
order, as train_depth.py:99-143 does.

Spans (round 3).  Whole batches quantise badly: 6 batches are 74 row tiles x 3 column tiles = 222 workgroups on 256 CUs (87 % of a
round), 7 batches are 261 (two rounds).  A forward may therefore cover a SPAN of the image stream that ends in the middle of a batch:
110 images = 21 670 rows = 85 x 3 = 255 tiles of the same round (tools/micro/fwd_rows.py: 10.18 ms for 96 images, 10.59 for 104,
10.99 for 110, 14.5 for 111); at 480x640 (1201 rows per image) 18 images instead of 16.  The engine keeps the tap-level rows of the
cut batch and completes it in the next span's forward (ViTEngine.forward_taps, ``Span``): the tap BN still sees exactly one batch's
16 images, so every batch still gets the bits it would get alone.  The carry (images of the cut batch already done) cycles through
B / gcd(T, B) values; one graph per (slot, carry) pattern is captured at the pipeline's first submit.
"""
from __future__ import annotations

import collections
import contextlib
import itertools
import os
import time
import weakref
from typing import Iterable, Iterator, Tuple

import torch

# a forward over a span of the image stream: images per batch, images of the cut batch carried in, id of the stream (= the pipeline's
# namespace: the engine keeps one carry store per stream, so two pipelines over one backbone — a training loop suspended at a cut batch
# and a validation pass — never see each other's rows)
Span = collections.namedtuple("Span", "batch carry stream", defaults=(0,))

_SLOT = 0  # slot of the forward being enqueued (host state; kernels are enqueued by one host thread)
_GROUPS = 1  # batches stacked into the forward being enqueued
_PIPELINED = False
_SHARED = False  # >= 3 kernel chains side by side: the GEMMs pick the tiles meant for a shared chip (mvp_hip.h, MVP_TILES_SHARED)
SHARED_TILES_FROM = 3
MAX_STREAMS = 3  # side streams = kernel chains running side by side; more than 3 measured slower (4: -14 %)


_NAMESPACES = itertools.count(1)  # one per FeaturePipeline: its slots' buffers are its own (two pipelines over one backbone do not collide)


def current_slot():
    """Key of the buffer set the forward being enqueued owns: 0 outside a pipeline, (pipeline namespace, slot index) inside."""
    return _SLOT


def current_groups():
    """Number of equal batches stacked along dim 0 of the forward being enqueued (1 outside a grouped pipeline forward), or the
    ``Span`` the forward covers."""
    return _GROUPS


def pipelined() -> bool:
    """True while a forward is being enqueued on a pipeline side stream."""
    return _PIPELINED


class GroupedFeatures(list):
    """Result of a grouped forward: the features of each batch of the group, in batch order (each entry is what ``model(images)``
    returns for that batch alone)."""


def tile_policy() -> int:
    """mvp_gemm_args.tile_policy of a GEMM launched now: 1 (MVP_TILES_SHARED) inside a forward of a pipeline that runs three kernel
    chains side by side, else 0 (MVP_TILES_ALONE)."""
    return 1 if _SHARED else 0


@contextlib.contextmanager
def _slot(i: int, chains: int = 2, groups: int = 1):
    global _SLOT, _PIPELINED, _SHARED, _GROUPS
    prev = (_SLOT, _PIPELINED, _SHARED, _GROUPS)
    _SLOT, _PIPELINED, _SHARED, _GROUPS = i, True, chains >= SHARED_TILES_FROM, groups
    try:
        yield
    finally:
        _SLOT, _PIPELINED, _SHARED, _GROUPS = prev


def publish() -> None:
    """Call once after building a long-lived device buffer that launches on ANY stream will read (split weights, resized
    pos-embed, constant pages): its initialising kernels ran on the current stream only, and a forward on another stream could
    otherwise read it early.  A device-wide sync, paid once per buffer."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()


_DEFERRED = []  # (batch of the group, update) registered by the forward being enqueued, applied by the consumer in batch order


def defer(fn, group: int = 0) -> None:
    """Register an update of state shared by all forwards (the tap-BN running statistics and step counter: the only state a frozen
    forward mutates).  A pipelined forward must not apply it itself — forwards on different streams finish in any order — so
    ``FeaturePipeline.next()`` runs it on the trainer's stream when the batch is handed over: batch order, no cross-stream event.
    ``group``: which batch of a grouped forward the update belongs to."""
    _DEFERRED.append((group, fn))


def _take_deferred():
    global _DEFERRED
    out, _DEFERRED = _DEFERRED, []
    return out


def freeze_gc() -> None:
    """Move everything allocated so far (torch's modules: ~10^6 objects) into the collector's permanent generation.  A generation-2
    collection otherwise walks all of it in the middle of the loop: a 75-110 ms host pause once per ~100 steps of this step, which
    drains the device queue (tools/micro/leg_timeline.py: 100-step legs at 2.44 ms/step without the pause, 2.78-3.08 with it)."""
    import gc

    gc.collect()
    gc.freeze()


@contextlib.contextmanager
def shared_tiles(on: bool = True):
    """Force the GEMM tile policy of launches made inside the block (measurement: the shared-chip tiles on a serial chain, so that a
    profiler sees the kernels of a pipelined run one at a time)."""
    global _SHARED
    prev = _SHARED
    _SHARED = bool(on)
    try:
        yield
    finally:
        _SHARED = prev


def default_depth(probe=None, group: int = 1) -> int:
    """Forwards submitted ahead of the probe step (= buffer slots) in the trainers and bench.py.  They run on min(depth, 3) side
    streams.  MVP_INFLIGHT wins when set (1 = everything on the trainer's stream).  Otherwise: 1 under a DPT probe whose forwards are
    single batches (``FeaturePipeline(ungrouped_depth=...)``: grouped ones do run beside it); 2 for grouped
    forwards (``group`` > 1: one group is being consumed batch by batch while the next one's forward runs); 4 single-batch forwards.
    Measured on MI355X at B=16 (bench.py, img/s), single-batch forwards: linear probe at 224^2 — one chain 5440-5660; two chains
    6380-6450; three chains 6610 with the same tiles and 7150-7340 with the shared-chip tiles (tile_policy); four chains 6210-6340, five
    6820-7030, six 6210-7120 (GPU_MAX_HW_QUEUES 8 / 16 change nothing).  Three chains with 4 / 5 / 6 slots — the next forward of a chain
    starts when the chain's previous one ends, without waiting for that batch's probe step — 7400-7530 / 7390-7570 / 7550; 2 chains + 4
    slots 6710-6800.  The DPT probe step (19 ms of chip-filling convolutions per batch) loses 739-748 -> 699-723 to a forward beside it."""
    env = os.environ.get("MVP_INFLIGHT")
    if env is not None:
        return max(1, int(env))
    if probe is not None and "_dpt_" in str(getattr(probe, "name", "")):
        return 1
    if os.environ.get("MVP_FORCE_DEVICE") is not None:
        # several ranks rehearsing on ONE card (tests, bench.py over gloo): their queues oversubscribe the card's hardware
        # queues and the processes get time-sliced (measured: 150 ms per step instead of 3)
        return 1
    return 2 if group > 1 else 4


def rows_per_image(H: int, W: int, P: int) -> int:
    """Token rows of one image, with the grid ViTEngine.tokens() builds: when either dimension is ragged against the patch size,
    center_padding pads BOTH by ``P - dim % P`` — a full extra patch for the dimension that was not ragged (utils.py:55-72)."""
    rh, rw = H % P, W % P
    ph, pw = (0, 0) if (rh == 0 and rw == 0) else (P - rh, P - rw)
    return 1 + ((H + ph) // P) * ((W + pw) // P)


GROUP_ROWS = 19000  # token rows a grouped forward aims at: 6 batches of 16 x 197 = 18912 rows = 74 x {3, 9, 12} tiles of 256^2, 87 % of whole rounds of 256 CUs
MAX_GROUP = 8


def default_group(model, images: torch.Tensor, depth: int) -> int:
    """Batches per grouped forward for batches shaped like ``images``.  MVP_PIPELINE_GROUP wins when set.  1 when the pipeline runs
    inline (depth 1) or the backbone cannot group; else as many batches as bring one forward to about GROUP_ROWS token rows."""
    env = os.environ.get("MVP_PIPELINE_GROUP")
    ok = getattr(model, "supports_grouping", None)
    if depth <= 1 or ok is None or not ok():
        return 1
    if env is not None:
        return max(1, min(MAX_GROUP, int(env)))
    P = int(getattr(model, "patch_size", 16))
    B, H, W = images.shape[0], images.shape[-2], images.shape[-1]
    rows = B * rows_per_image(H, W, P)
    return max(1, min(MAX_GROUP, int(round(GROUP_ROWS / rows))))


MAX_SPAN_PATTERNS = 16  # (slot, carry) patterns whose graphs are captured up front; a span length with more runs eagerly


def default_span(model, images: torch.Tensor, depth: int, group: int) -> int:
    """IMAGES per forward when forwards may end in the middle of a batch (0: whole batches only).  MVP_PIPELINE_SPAN wins when set.
    As many images as keep the narrowest GEMM of a block (N = C columns: ceil(C / 256) column tiles of the 256x256 kernel) within
    one round of the device's CUs (256 on MI355X; mvp_info().cu_count), rounded down to a multiple of B / 8 (few carry patterns); 0 when that is a whole number of batches anyway
    (then ``group`` covers it), less than one batch, the pipeline is not the two-slot one, or a single batch already fills a forward
    (``group`` 1, e.g. B = 16 at 480x640: 18 images per forward on one stream measured 1293 img/s against 1349 for single batches on
    three streams, whose chains fill each other's partial rounds; at 224^2, B = 16 / 64: 8967 / 9203 against 8514 / 8191)."""
    env = os.environ.get("MVP_PIPELINE_SPAN")
    B = images.shape[0]
    ok = getattr(model, "supports_grouping", None)
    if depth != 2 or ok is None or not ok() or (env is not None and int(env) <= 0):
        return 0
    if env is not None:
        T = int(env)
    elif group <= 1:
        return 0
    else:
        eng = model.engine() if hasattr(model, "engine") else None
        C = int(getattr(eng, "C", 0) or 0)
        if C <= 0:
            return 0
        P = int(getattr(model, "patch_size", 16))
        H, W = images.shape[-2], images.shape[-1]
        rows = rows_per_image(H, W, P)
        unit = max(1, B // 8)
        T = min(((_cu_count() // (-(-C // 256))) * 256) // rows, MAX_GROUP * B) // unit * unit
    return T if (T > B and T % B) else 0


def _cu_count() -> int:
    """Compute units of the current device (one 256x256 tile of the large-M GEMM per CU and round); 256 when the library cannot say."""
    try:
        from . import lib

        n = int(lib.info().cu_count)
        return n if n > 0 else 256
    except Exception:
        return 256


def span_patterns(T: int, B: int):
    """(slot, carry) of consecutive span forwards of T images from a batch boundary, one full cycle (the slots alternate)."""
    out, k = [], 0
    while True:
        out.append((k % 2, (k * T) % B))
        k += 1
        if (k * T) % B == 0 and k % 2 == 0:
            return out


def _tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _tensors(o)
        for name in ("cls", "stats"):  # TapOutputs side products allocated by the forward
            extra = getattr(obj, name, None)
            if extra is not None and extra is not obj:
                yield from _tensors(extra)


def _multi_rank() -> bool:
    import torch.distributed as dist

    return bool(dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)


class FeaturePipeline:
    """``submit(images)`` enqueues ``model(images)`` on a side stream; ``next()`` returns the oldest submitted features on the
    caller's current stream (event wait, no host sync) after applying the forward's deferred state updates.  ``depth`` = forwards
    in flight; backbones without ``supports_pipelining`` run inline on the caller's stream, as does depth 1.
    ``submit_group([images, ...])`` enqueues ONE forward over several equal-shaped batches (see the module docstring); ``next()``
    still hands the batches over one at a time, in order.  ``group`` = batches per full group (1: single-batch forwards; None:
    ``default_group`` at the first submit, what ``pipelined_features`` and bench.py ask for); ``free_slots()`` tells how many more
    forwards may be submitted.

    ``graphs`` (default: MVP_PIPELINE_GRAPHS when set, else on unless the job has more than one rank): a ``graph_safe`` backbone's forward is captured once per (slot, input shape) in a
    hipGraph and replayed — one launch call instead of ~100, 1.4 ms of host time per step down to 0.4, so a busy host (data loading,
    logging) no longer starves the device.  The first forward of every slot runs eagerly (it allocates the slot's buffers and builds
    lazily cached operands) and is then captured; later ones replay; the input batch is copied into the graph's static buffer.
    Of grouped forwards only FULL groups are captured (all slots at the pipeline's first submit, whatever that submit's size, so
    that no capture falls into the run); a ragged last group runs eagerly.
    A graph is tied to the engine it was captured from (rebuilt weights invalidate it: new key, new capture)."""

    def __init__(self, model, depth: int = None, run_ahead: int = None, graphs: bool = None, streams: int = None, group: int = 1, span: int = None,
                 ungrouped_depth: int = None):
        """``run_ahead``: the host may be at most this many forwards ahead of the device (MVP_RUN_AHEAD, default 8; 0 = unbounded).
        The reference's loop syncs every step (``loss.item()``, train_depth.py:143); a loop that never syncs would otherwise queue
        hundreds of launches (and keep their argument buffers alive).  Throughput-neutral on MI355X (tools/micro/pipeline_probe.py,
        B=16, 300 steps, 2 in flight: 6526 img/s unbounded, 6565 with 8, 6584 with 3)."""
        self._depth_arg = depth
        depth = default_depth() if depth is None else int(depth)
        if depth < 1:
            raise ValueError("depth must be >= 1")
        if not getattr(model, "supports_pipelining", False):
            depth = 1
        self.model, self.depth = model, depth
        self.group = None if group is None else max(1, int(group))  # resolved at the first submit (needs the batch shape)
        self._group_arg, self._resolved_for = self.group, None
        if self.depth == 1:
            self.group = 1
        # images per span forward (module docstring, "Spans"): None = default_span when ``group`` is chosen automatically, else off
        self._ns = next(_NAMESPACES)
        self._lent = False  # pipelined_features: this (cached) pipeline is driving a loop right now
        self._span_arg = span
        # depth to fall to when ``depth`` is None and the forwards turn out to be single batches (``default_depth(probe)``: a DPT probe
        # step gains from grouped forwards beside it, 762 -> 802 img/s, but loses to single-batch ones, 739-748 -> 699-723)
        self._ungrouped_depth = ungrouped_depth
        self.span, self.span_batch, self._span_resolved = 0, 0, False
        # (stream priorities do not help: the device offers only (0, -1), and high-priority side streams measured the same)
        # ``depth`` forwards are submitted ahead (one buffer slot each); they run on ``streams`` side streams = kernel chains side by side
        self._streams_given = streams is not None or os.environ.get("MVP_PIPELINE_STREAMS") is not None
        if streams is None:
            streams = int(os.environ.get("MVP_PIPELINE_STREAMS", str(MAX_STREAMS)))
        self._streams_arg = int(streams)
        self.chains = max(1, min(depth, int(streams)))
        self.streams = [torch.cuda.Stream() for _ in range(self.chains)] if depth > 1 else []
        self._queue = collections.deque()  # one entry per BATCH: (features, completion event of its forward, deferred updates)
        self._open = collections.deque()   # per forward in flight: [batches of it not yet handed over, its slot]
        self._n = 0
        self.run_ahead = int(os.environ.get("MVP_RUN_AHEAD", "8")) if run_ahead is None else int(run_ahead)
        self._issued = collections.deque()  # completion events of the newest ``run_ahead`` forwards
        if graphs is None:
            # MVP_PIPELINE_GRAPHS wins when set.  Otherwise ON for single-process jobs and OFF (eager launches) for jobs with more than one
            # rank: capture and replay beside RCCL's stream and watchdog thread have run on this pool only with ONE rank (a real RCCL
            # all-reduce per step at world size 1: tests/test_gpu_dist.py::test_rccl_world1_*), never beside a collective that waits for
            # peers — and a lazy capture (an epoch's ragged last batch, a rebind) is a device-wide sync that could land while the previous
            # step's all-reduce is still pending.  With span forwards the replay saves ~100 launch calls per 110 images (~15 per probe
            # step, against the ~115 the probe step itself makes), so the eager default costs multi-rank jobs little.
            env = os.environ.get("MVP_PIPELINE_GRAPHS")
            graphs = (env != "0") if env is not None else not _multi_rank()
        self.graphs = bool(graphs) and depth > 1 and bool(getattr(model, "graph_safe", False))
        self._warned_eager_span = False
        self.throttle_wait_s = 0.0  # wall time the host has spent waiting in the ``run_ahead`` throttle (not work: the device was behind)
        self._graphs = {}  # (slot, shape, dtype, training, engine id, group) -> dict(calls, graph, static_in, feats, deferred)
        self._stage = {}   # (slot, shape, dtype) -> stacked input buffer of eager grouped forwards
        self._static = {}  # (slot, shape, dtype) -> static input buffer of that slot's graphs

    def __len__(self) -> int:
        """Batches submitted and not yet handed over."""
        return len(self._queue)

    def free_slots(self) -> int:
        """Forwards that may still be submitted before ``next()`` has to be called."""
        return self.depth - len(self._open)

    def resolve_group(self, images: torch.Tensor) -> int:
        """Fix the shape of the forwards at the first submit (needs the batch shape): batches per group, images per span, slots, streams."""
        if self._span_resolved:
            return self.group
        self._span_resolved = True
        self._resolved_for = (tuple(images.shape), images.dtype)
        auto = self.group is None
        depth_free = self._depth_arg is None and os.environ.get("MVP_INFLIGHT") is None
        B = images.shape[0]
        if auto:
            self.group = default_group(self.model, images, self.depth)
        span = 0
        if self.depth > 1 and (self.depth == 2 or depth_free) and not (self._streams_given and self._streams_arg != 1):
            if self._span_arg is not None:
                span = int(self._span_arg) if (int(self._span_arg) > B and int(self._span_arg) % B) else 0
            elif auto:
                span = default_span(self.model, images, 2, self.group)
        if span:
            # span forwards: two slots, ONE side stream (it orders the carry store between consecutive forwards)
            self.span, self.span_batch = span, B
            self.depth, self.chains = 2, 1
            self.streams = self.streams[:1] if self.streams else [torch.cuda.Stream()]
            self.group = max(self.group, -(-span // B))
        elif auto and self.group > 1:
            if depth_free:
                # grouped forwards: two slots (one group being consumed, the next one's forward running); see default_depth
                self.depth = 2
            # ... on ONE side stream unless asked otherwise: a grouped forward's GEMMs fill the chip by themselves (one 256x256 tile per
            # CU), a second forward chain beside them adds nothing (measured, B = 16, 6 batches per forward, img/s at 20 / 60 steps:
            # one stream 8357 / 8706, two 8326 / 8714), and one chain keeps every per-kernel measurement in the regime of the timed run
            self.chains = max(1, min(self.depth, self._streams_arg if self._streams_given else 1))
            self.streams = self.streams[:self.chains]
        elif auto and depth_free and self._ungrouped_depth is not None and self._ungrouped_depth < self.depth:
            self.depth = max(1, int(self._ungrouped_depth))
            self.chains = max(1, min(self.depth, self.chains))
            self.streams = self.streams[:self.chains] if self.depth > 1 else []
        return self.group

    def rebind(self, images: torch.Tensor) -> None:
        """A cached pipeline (pipelined_features keeps one per model and mode) meets batches of another shape: choose group / span /
        slots again for that shape.  Only while nothing is in flight.  The graphs already captured stay (they are keyed by shape); the
        new shape's span forwards run eagerly unless this pipeline has captured nothing yet."""
        if self._resolved_for is None or self._resolved_for == (tuple(images.shape), images.dtype):
            return
        if self._queue or self._open:
            raise RuntimeError("rebind() needs an empty pipeline")
        depth = default_depth() if self._depth_arg is None else int(self._depth_arg)
        if not getattr(self.model, "supports_pipelining", False):
            depth = 1
        self.depth, self.group = depth, (1 if depth == 1 else self._group_arg)
        self.chains = max(1, min(depth, self._streams_arg))
        while len(self.streams) < (self.chains if depth > 1 else 0):
            self.streams.append(torch.cuda.Stream())
        self.streams = self.streams[:self.chains] if depth > 1 else []
        self.span, self.span_batch, self._span_resolved, self._n = 0, 0, False, 0
        self.resolve_group(images)

    # ------------------------------------------------------------------ one forward on a slot's stream
    def _eager(self, slot: int, images: torch.Tensor, groups: int = 1):
        with _slot((self._ns, slot), self.chains, groups):
            _take_deferred()
            feats = _extract(self.model, images)
            return feats, _take_deferred()

    def _forward(self, slot: int, s, batches, G):
        """Runs on stream ``s`` (current).  ``batches``: the image tensors of this forward, stacked along dim 0 — G whole batches
        (``G`` an int) or the pieces of a span (``G`` a ``Span``).  Returns (features, deferred updates): for G != 1 a GroupedFeatures
        and (batch index, update) pairs covering all the batches the forward completes."""
        first = batches[0]
        is_span = isinstance(G, Span)
        shape = (sum(b.shape[0] for b in batches),) + tuple(first.shape[1:])
        eng = self.model.engine() if hasattr(self.model, "engine") else None
        key = (slot, shape, first.dtype, bool(self.model.training), id(eng), G)
        if is_span:
            # span forwards replay the graph of their (slot, carry) pattern when ``_precapture_spans`` set one up at the pipeline's first
            # submit; anything else (a short span at the stream's end, a stream that restarted out of phase) runs eagerly rather than
            # pay a capture — a device-wide sync — inside the run
            ent = self._graphs.get(key) if self.graphs else None
            if ent is None:
                if self.graphs and self._graphs and shape[0] == self.span and not self._warned_eager_span:
                    self._warned_eager_span = True  # a FULL span without a graph: worth a word (short spans at a stream's end are expected)
                    import warnings

                    warnings.warn(f"mvp.pipeline: no captured graph for span pattern {G} on slot {slot}; this forward runs eagerly")
                return self._eager(slot, self._stacked(slot, batches, shape), G)
            ent["calls"] += 1
        else:
            full = G == self.group or G == 1 and self.group in (None, 1)
            if not (self.graphs and full):
                return self._eager(slot, self._stacked(slot, batches, shape) if G > 1 else first, G)
            ent = self._graphs.get(key)
            if ent is None:
                # at most two whole-batch shapes per slot (full batches + an epoch's ragged last one); the span patterns captured up
                # front are not part of that budget (they are never re-captured: evicting them would silently turn those forwards eager)
                mine = [k for k in self._graphs if k[0] == slot and not isinstance(k[5], Span)]
                for k in mine[:-1] if len(mine) >= 2 else []:
                    del self._graphs[k]
                ent = self._graphs[key] = dict(calls=0, graph=None)
            else:
                self._graphs[key] = self._graphs.pop(key)  # most recently used last
            ent["calls"] += 1
            if ent["graph"] is None and not any(e["graph"] is not None for e in self._graphs.values()):
                # The very first forward of the pipeline: set up EVERY slot now (the set-up forwards write nothing but the slot's own
                # buffers, their deferred updates are dropped), so that no capture — each one is a device-wide sync — falls into the run
                # later, whatever the caller's warm-up length.  Later shapes (an epoch's ragged last batch) are set up lazily per slot.
                for other in range(self.depth):
                    if other != slot:
                        okey = (other,) + key[1:]
                        oent = self._graphs[okey] = dict(calls=0, graph=None)
                        self._capture(other, s, batches, shape, G, oent, eng)
            if ent["graph"] is None:
                return self._capture(slot, s, batches, shape, G, ent, eng)  # (its replay computed this forward's features)
        self._fill(ent["static_in"], batches, G)
        ent["graph"].replay()
        from .vit import register_pack

        for pk in ent["packs"]:
            pk.generation += 1  # the host code that counts rewrites of the packing does not run on a replay ...
            register_pack(pk.source_refs, pk)  # ... nor does the registration: an entry aged out of the registry (other pipelines' packings) comes back
        return ent["feats"], ent["deferred"]

    def _precapture_spans(self, s, sample: torch.Tensor) -> None:
        """Capture the graph of every (slot, carry) pattern full spans cycle through, on copies of ``sample`` (results and deferred
        updates dropped; the carry store they scribble on is not read before a real forward has written it)."""
        T, B = self.span, self.span_batch
        pats = span_patterns(T, B)
        if len(pats) > MAX_SPAN_PATTERNS:
            return
        filler = []
        while sum(f.shape[0] for f in filler) < T:
            filler.append(sample[:T - sum(f.shape[0] for f in filler)])
        shape = (T,) + tuple(sample.shape[1:])
        eng = self.model.engine() if hasattr(self.model, "engine") else None
        for slot, carry in pats:
            G = Span(B, carry, self._ns)
            ent = self._graphs[(slot, shape, sample.dtype, bool(self.model.training), id(eng), G)] = dict(calls=0, graph=None)
            self._capture(slot, s, filler, shape, G, ent, eng)

    @staticmethod
    def _fill(dst: torch.Tensor, batches, G=None) -> None:
        at = 0
        for b in batches:
            dst[at:at + b.shape[0]].copy_(b, non_blocking=True)
            at += b.shape[0]

    def _stacked(self, slot: int, batches, shape) -> torch.Tensor:
        """The [G * B, ...] input of an eager grouped forward: a per-slot staging buffer (reused, so the steady state allocates nothing)."""
        key = (slot, shape, batches[0].dtype)
        buf = self._stage.get(key)
        if buf is None:
            self._stage = {k: v for k, v in self._stage.items() if k[0] != slot}
            buf = self._stage[key] = torch.empty(shape, dtype=batches[0].dtype, device=batches[0].device)
        self._fill(buf, batches, len(batches))
        return buf

    def _capture(self, slot: int, s, batches, shape, G: int, ent: dict, eng):
        """Set a slot up for this input shape: an eager forward (allocates the slot's buffers, builds lazily cached operands — none of
        that may happen inside a capture), the capture of the very same call, and a first replay on the real input (a graph's first launch
        uploads it to the device: that cost belongs here, not in the run).  Returns the replay's (features, deferred updates)."""
        from .vit import lookup_pack

        skey = (slot, tuple(shape), batches[0].dtype)  # one static input per slot and shape (a slot's span patterns share it)
        static_in = self._static.get(skey)
        if static_in is None:
            static_in = self._static[skey] = torch.empty(shape, dtype=batches[0].dtype, device=batches[0].device)
        self._fill(static_in, batches, G)
        self._eager(slot, static_in, G)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):  # other threads (allocator, collectives' watchdog) stay free
            feats, deferred = self._eager(slot, static_in, G)
        per_batch = list(feats) if isinstance(feats, GroupedFeatures) else [feats]
        packs = [pk for pk in (lookup_pack(f) if isinstance(f, (list, tuple)) else None for f in per_batch) if pk is not None]
        ent.update(graph=g, static_in=static_in, feats=feats, deferred=deferred, packs=packs,
                   # the graph holds raw addresses of the slot's buffers and of the engine's operands: both stay alive with it
                   keep=(eng, eng.slot_state((self._ns, slot)) if hasattr(eng, "slot_state") else None))
        g.replay()
        return feats, deferred

    def submit(self, images: torch.Tensor) -> None:
        self.submit_group([images])

    def submit_group(self, batches) -> None:
        """Enqueue ONE forward over ``batches`` (a list of equal-shaped image batches; a single one = the plain forward)."""
        batches = list(batches)
        G = len(batches)
        if G < 1:
            raise ValueError("empty group")
        self.resolve_group(batches[0])
        if G > 1 and (any(b.shape != batches[0].shape or b.dtype != batches[0].dtype for b in batches) or self.depth == 1):
            raise ValueError("a grouped forward needs equal-shaped batches and a pipeline of depth >= 2")
        self._submit(batches, G, G)

    def submit_span(self, pieces, batch: int, carry: int) -> None:
        """Enqueue ONE forward over a span of the image stream (module docstring, "Spans"): ``pieces`` = image tensors, consecutive in
        the stream, whose first ``batch - carry`` images (when carry > 0) complete the batch the previous span cut; the span may end
        inside a batch again.  ``next()`` hands over the (carry + images) // batch batches the forward completes, in order."""
        pieces = list(pieces)
        if not pieces or self.depth != 2 or self.chains != 1:
            raise ValueError("a span forward needs images and the two-slot, one-stream pipeline of grouped forwards")
        if any(p.shape[1:] != pieces[0].shape[1:] or p.dtype != pieces[0].dtype for p in pieces):
            raise ValueError("a span forward needs equal-shaped images")
        T = sum(p.shape[0] for p in pieces)
        if not 0 <= carry < batch or (carry + T) // batch < 1:
            raise ValueError(f"span of {T} images with carry {carry} completes no batch of {batch}")
        self._submit(pieces, Span(int(batch), int(carry), self._ns), (carry + T) // batch)

    def _submit(self, batches, G, nb: int) -> None:
        if self.free_slots() <= 0:
            raise RuntimeError(f"{self.depth} forwards already in flight: call next() first")
        if self.run_ahead > 0 and len(self._issued) >= self.run_ahead:
            t0 = time.perf_counter()
            self._issued.popleft().synchronize()  # host waits for the forward issued ``run_ahead`` submissions ago
            self.throttle_wait_s += time.perf_counter() - t0  # (bench.py tells host WORK from this wait)
        if self.depth == 1:
            feats = _extract(self.model, batches[0])
            if self.run_ahead > 0 and batches[0].is_cuda:
                ev = torch.cuda.Event()
                ev.record()
                self._issued.append(ev)
            self._queue.append((feats, None, ()))
            self._open.append([1, 0])
            return
        is_span = isinstance(G, Span)
        if is_span and not self._open and G.carry == 0:
            self._n = 0  # a stream that (re)starts on a batch boundary starts the (slot, carry) cycle of the captured graphs over
        # slots rotate — but never a slot whose previous forward still has batches to hand over
        busy = {e[1] for e in self._open}
        slot = self._n % self.depth
        if slot in busy:
            slot = next(i for i in range(self.depth) if i not in busy)
        s = self.streams[self._n % len(self.streams)]
        self._n += 1
        cur = torch.cuda.current_stream()
        # the batches are ready on the caller's stream, and the probe steps that read this slot's buffers are already enqueued there
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            if self.graphs and not self._graphs:
                if is_span and self.span:
                    # the pipeline's first forward: set the graphs of all full-span patterns up NOW, whatever this forward's length (a
                    # warm-up is shorter than a span), so that no capture falls into the run later
                    self._precapture_spans(s, batches[0])
                elif not is_span and self.group > 1 and G != self.group:
                    # the same for a ragged first group
                    self._forward(slot, s, [batches[0]] * self.group, self.group)
            feats, deferred = self._forward(slot, s, batches, G)
            done = torch.cuda.Event()
            done.record(s)
        if self.run_ahead > 0:
            self._issued.append(done)
        for b in batches:
            if b.is_cuda:
                b.record_stream(s)  # allocated on the caller's stream, read on the side stream
        if G == 1 and not isinstance(feats, GroupedFeatures):
            self._queue.append((feats, done, [fn for _, fn in deferred]))
        else:
            if not isinstance(feats, GroupedFeatures) or len(feats) != nb:
                raise RuntimeError("the backbone did not return one result per batch of the group")
            for g in range(nb):
                self._queue.append((feats[g], done, [fn for gg, fn in deferred if gg == g]))
        self._open.append([nb, slot])

    def next(self):
        feats, done, deferred = self._queue.popleft()
        self._open[0][0] -= 1
        if self._open[0][0] == 0:
            self._open.popleft()
        if done is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(done)
            for t in _tensors(feats):
                t.record_stream(cur)  # allocated on the side stream, read (and later freed) under the caller's stream
            for fn in deferred:  # tap-BN running statistics: applied here, on the caller's stream, in batch order
                fn()
        return feats

    def drain(self) -> None:
        """Discard the forwards still in flight (a loop left early): the caller's stream waits for them, their deferred state updates
        are NOT applied — as if those batches had never been run, which is what the one-batch-at-a-time loop would have done."""
        while self._queue:
            feats, done, deferred = self._queue.popleft()
            if done is not None:
                torch.cuda.current_stream().wait_event(done)
        self._open.clear()


def _extract(model, images):
    from .train import extract_features

    return extract_features(model, images)


_PIPELINES = weakref.WeakKeyDictionary()  # model -> {(training, depth, group, hint): FeaturePipeline}
_NO_CACHE: dict = {}  # what cached_pipelines returns for a model that cannot be weakly referenced: always empty, nothing is kept


def cached_pipelines(model) -> dict:
    """The pipelines ``pipelined_features`` keeps for ``model`` (created on demand; empty for objects that cannot be weakly referenced)."""
    try:
        return _PIPELINES.setdefault(model, {})
    except TypeError:
        return _NO_CACHE


def pipelined_features(model, batches: Iterable, image_key="image", depth: int = None, probe=None, group: int = None,
                       pipe: "FeaturePipeline" = None, graphs: bool = None) -> Iterator[Tuple[object, object]]:
    """Yield ``(batch, features)`` for every batch of ``batches`` with up to ``depth`` forwards in flight: the forwards of the next
    batches are enqueued before batch t is handed to the caller, so they run under the caller's probe step t.  ``depth`` None:
    ``default_depth(probe)``.  Consecutive equal-shaped batches are stacked ``group`` at a time into one forward (None:
    ``default_group``; an epoch's ragged last batch and whatever is left over run as smaller forwards).  ``pipe``: an existing
    pipeline to reuse (its captured graphs); it must be empty.  ``graphs``: handed to ``FeaturePipeline`` (None: its default — off for
    jobs with more than one rank; a loop with NO collective in flight, e.g. the sharded SPair evaluation, passes True)."""
    if pipe is None:
        hint = None
        if depth is None:
            hint = default_depth(probe)
            # MVP_INFLIGHT and ranks rehearsing on one card are final; otherwise the pipeline picks its own default once it knows
            # whether the forwards are grouped (2 slots, DPT probe included) or single batches (4; 1 under a DPT probe)
            depth = hint if (os.environ.get("MVP_INFLIGHT") is not None or os.environ.get("MVP_FORCE_DEVICE") is not None) else None
        # One pipeline per (model, mode, shape of the request), kept beside the model (a weak map: nothing is attached to the module, so
        # deepcopy / pickling of the model are unaffected, and the pipelines go when the model goes): every epoch's loop and every validation pass replay the
        # graphs the first one captured (setting a pipeline's graphs up costs ~0.25 s: a tenth of an NYU-sized epoch at this throughput).
        # A cached pipeline that is still in use (a loop suspended mid-epoch) is left alone: the caller gets a fresh one.
        cache = cached_pipelines(model)
        key = (bool(getattr(model, "training", False)), depth, group, hint) + (() if graphs is None else (bool(graphs),))
        pipe = cache.get(key)
        if pipe is None or len(pipe) or pipe._open or pipe._lent:
            # (the cached pipeline refers to its model weakly: the map's values must not keep its keys alive)
            keep = cache is not _NO_CACHE
            pipe = FeaturePipeline(weakref.proxy(model) if keep else model, depth, group=group, ungrouped_depth=hint, graphs=graphs)
            if keep:
                cache[key] = pipe
        pipe._lent = True
        lent = pipe
    elif len(pipe):
        raise RuntimeError("pipelined_features needs an empty pipeline")
    else:
        lent = None
    if hasattr(batches, "consumer_lag"):
        # mvp.prefetch.DevicePrefetcher recycles its device buffers (it sizes its pool when its first batch is pulled): this generator
        # holds up to depth x group batches before the caller has issued the probe step of the first of them (which reads that batch's
        # target; its image is read by the forward in flight).  The group size is not known before the first batch: its upper bound.
        # (a span forward also holds the batch it cuts)
        g_hold = (pipe.group or MAX_GROUP) + (1 if (pipe.span or pipe.group is None) else 0)
        if lent is not None and pipe._resolved_for is not None:
            g_hold = max(g_hold, MAX_GROUP + 1)  # a cached pipeline may be re-shaped by this loader's first batch (rebind)
        batches.consumer_lag = max(int(batches.consumer_lag), pipe.depth * g_hold - 1)
    it = iter(batches)
    pending = collections.deque()
    held = []  # a batch pulled from the iterator that did not fit the group being formed

    def images_of(b):
        x = b[image_key] if isinstance(b, dict) else b[0]
        if x.is_cuda or not torch.cuda.is_available():  # (a host tensor without a device: the backbone raises MvpError)
            return x
        return x.to(torch.device("cuda", torch.cuda.current_device()), non_blocking=True)

    def pull():
        if held:
            return held.pop()
        try:
            b = next(it)
        except StopIteration:
            return None
        return b, images_of(b)

    cut = []  # [(batch, images, images of it already forwarded)]: the batch the previous span forward ended in

    def feed_span() -> bool:
        """One forward over the next ``pipe.span`` images of the stream (fewer at its end or where the batch shape changes)."""
        B = pipe.span_batch
        pieces, done, carry, room = [], [], 0, pipe.span
        if cut:
            b, x, used = cut.pop()
            pieces.append(x[used:])  # (a span is longer than a batch: the cut batch always completes here)
            done.append(b)
            carry, room = used, room - (B - used)
        while room > 0:
            nxt = pull()
            if nxt is None:
                break
            ref = pieces[0] if pieces else None
            if nxt[1].shape[0] != B or (ref is not None and (nxt[1].shape[1:] != ref.shape[1:] or nxt[1].dtype != ref.dtype)):
                held.append(nxt)  # (the span ends on a batch boundary here: a batch is only ever cut as the last piece)
                break
            if room >= B:
                pieces.append(nxt[1])
                done.append(nxt[0])
                room -= B
            else:
                pieces.append(nxt[1][:room])
                cut.append((nxt[0], nxt[1], room))
                room = 0
        if not pieces:
            return False
        if not done:  # part of one batch and nothing else (cannot happen: spans are longer than a batch): run the batch whole
            b, x, _ = cut.pop()
            pipe.submit_group([x])
            pending.append(b)
            return True
        pipe.submit_span(pieces, B, carry)
        pending.extend(done)
        return True

    started = []

    def feed() -> bool:
        if cut:
            return feed_span()
        first = pull()
        if first is None:
            return False
        if not started:  # a pipeline taken from the model's cache may have been shaped for other batches
            started.append(True)
            if lent is not None:
                pipe.rebind(first[1])
        G = pipe.resolve_group(first[1])
        if pipe.span and first[1].shape[0] == pipe.span_batch:
            held.append(first)
            return feed_span()
        grp = [first]
        while len(grp) < G:
            nxt = pull()
            if nxt is None:
                break
            if nxt[1].shape != first[1].shape or nxt[1].dtype != first[1].dtype:
                held.append(nxt)
                break
            grp.append(nxt)
        pipe.submit_group([x for _, x in grp])
        pending.extend(b for b, _ in grp)
        return True

    try:
        while True:
            while pipe.free_slots() > 0 and feed():
                pass
            if not pending:
                break
            b = pending.popleft()
            yield b, pipe.next()
    finally:
        pipe.drain()
        if lent is not None:
            lent._lent = False  # (the model's cached pipeline is free for the next loop)
