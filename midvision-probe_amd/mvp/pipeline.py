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
"""
from __future__ import annotations

import collections
import contextlib
import os
from typing import Iterable, Iterator, Tuple

import torch

_SLOT = 0  # slot of the forward being enqueued (host state; kernels are enqueued by one host thread)
_PIPELINED = False
_SHARED = False  # >= 3 kernel chains side by side: the GEMMs pick the tiles meant for a shared chip (mvp_hip.h, MVP_TILES_SHARED)
SHARED_TILES_FROM = 3
MAX_STREAMS = 3  # side streams = kernel chains running side by side; more than 3 measured slower (4: -14 %)


def current_slot() -> int:
    return _SLOT


def pipelined() -> bool:
    """True while a forward is being enqueued on a pipeline side stream."""
    return _PIPELINED


def tile_policy() -> int:
    """mvp_gemm_args.tile_policy of a GEMM launched now: 1 (MVP_TILES_SHARED) inside a forward of a pipeline that runs three kernel
    chains side by side, else 0 (MVP_TILES_ALONE)."""
    return 1 if _SHARED else 0


@contextlib.contextmanager
def _slot(i: int, chains: int = 2):
    global _SLOT, _PIPELINED, _SHARED
    prev = (_SLOT, _PIPELINED, _SHARED)
    _SLOT, _PIPELINED, _SHARED = i, True, chains >= SHARED_TILES_FROM
    try:
        yield
    finally:
        _SLOT, _PIPELINED, _SHARED = prev


def publish() -> None:
    """Call once after building a long-lived device buffer that launches on ANY stream will read (split weights, resized
    pos-embed, constant pages): its initialising kernels ran on the current stream only, and a forward on another stream could
    otherwise read it early.  A device-wide sync, paid once per buffer."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()


_DEFERRED = []  # state updates registered by the forward being enqueued, applied by the consumer in batch order


def defer(fn) -> None:
    """Register an update of state shared by all forwards (the tap-BN running statistics and step counter: the only state a frozen
    forward mutates).  A pipelined forward must not apply it itself — forwards on different streams finish in any order — so
    ``FeaturePipeline.next()`` runs it on the trainer's stream when the batch is handed over: batch order, no cross-stream event."""
    _DEFERRED.append(fn)


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


def default_depth(probe=None) -> int:
    """Batches whose forward is submitted ahead of the probe step (= buffer slots) in the trainers and bench.py.  They run on
    min(depth, 3) side streams.  MVP_INFLIGHT wins when set (1 = everything on the trainer's stream).  Otherwise 4, except under a DPT
    probe.  Measured on MI355X at B=16 (bench.py, img/s): linear probe at 224^2 — one chain 5440-5660; two chains 6380-6450; three
    chains 6610 with the same tiles and 7150-7340 with the shared-chip tiles (tile_policy); four chains 6210-6340, five 6820-7030, six
    6210-7120 (GPU_MAX_HW_QUEUES 8 / 16 change nothing).  Three chains with 4 / 5 / 6 slots — the next forward of a chain starts when
    the chain's previous one ends, without waiting for that batch's probe step — 7400-7530 / 7390-7570 / 7550; 2 chains + 4 slots
    6710-6800.  The DPT probe step (19 ms of chip-filling convolutions per batch) loses 739-748 -> 699-723 to a forward beside it."""
    env = os.environ.get("MVP_INFLIGHT")
    if env is not None:
        return max(1, int(env))
    if probe is not None and "_dpt_" in str(getattr(probe, "name", "")):
        return 1
    if os.environ.get("MVP_FORCE_DEVICE") is not None:
        # several ranks rehearsing on ONE card (tests, bench.py over gloo): their queues oversubscribe the card's hardware
        # queues and the processes get time-sliced (measured: 150 ms per step instead of 3)
        return 1
    return 4


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


class FeaturePipeline:
    """``submit(images)`` enqueues ``model(images)`` on a side stream; ``next()`` returns the oldest submitted features on the
    caller's current stream (event wait, no host sync) after applying the forward's deferred state updates.  ``depth`` = forwards
    in flight; backbones without ``supports_pipelining`` run inline on the caller's stream, as does depth 1.

    ``graphs`` (default MVP_PIPELINE_GRAPHS != "0"): a ``graph_safe`` backbone's forward is captured once per (slot, input shape) in a
    hipGraph and replayed — one launch call instead of ~100, 1.4 ms of host time per step down to 0.4, so a busy host (data loading,
    logging) no longer starves the device.  The first forward of every slot runs eagerly (it allocates the slot's buffers and builds
    lazily cached operands) and is then captured; later ones replay; the input batch is copied into the graph's static buffer.
    A graph is tied to the engine it was captured from (rebuilt weights invalidate it: new key, new capture)."""

    def __init__(self, model, depth: int = None, run_ahead: int = None, graphs: bool = None, streams: int = None):
        """``run_ahead``: the host may be at most this many forwards ahead of the device (MVP_RUN_AHEAD, default 8; 0 = unbounded).
        The reference's loop syncs every step (``loss.item()``, train_depth.py:143); a loop that never syncs would otherwise queue
        hundreds of launches (and keep their argument buffers alive).  Throughput-neutral on MI355X (tools/micro/pipeline_probe.py,
        B=16, 300 steps, 2 in flight: 6526 img/s unbounded, 6565 with 8, 6584 with 3)."""
        depth = default_depth() if depth is None else int(depth)
        if depth < 1:
            raise ValueError("depth must be >= 1")
        if not getattr(model, "supports_pipelining", False):
            depth = 1
        self.model, self.depth = model, depth
        # (stream priorities do not help: the device offers only (0, -1), and high-priority side streams measured the same)
        # ``depth`` batches are submitted ahead (one buffer slot each); they run on ``streams`` side streams = kernel chains side by side
        if streams is None:
            streams = int(os.environ.get("MVP_PIPELINE_STREAMS", str(MAX_STREAMS)))
        self.chains = max(1, min(depth, int(streams)))
        self.streams = [torch.cuda.Stream() for _ in range(self.chains)] if depth > 1 else []
        self._queue = collections.deque()
        self._n = 0
        self.run_ahead = int(os.environ.get("MVP_RUN_AHEAD", "8")) if run_ahead is None else int(run_ahead)
        self._issued = collections.deque()  # completion events of the newest ``run_ahead`` forwards
        if graphs is None:
            env = os.environ.get("MVP_PIPELINE_GRAPHS")
            if env is not None:
                graphs = env != "0"
            else:
                # Capture with RCCL work in flight (its watchdog thread queries events) could not be exercised on the one-GPU test pool:
                # multi-rank jobs launch eagerly unless asked otherwise.  At the per-GPU batch sizes where replay pays (B <= 8) set
                # MVP_PIPELINE_GRAPHS=1; at B = 16 the run is device-bound either way.
                import torch.distributed as dist

                graphs = not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        self.graphs = bool(graphs) and depth > 1 and bool(getattr(model, "graph_safe", False))
        self._graphs = {}  # (slot, shape, dtype, training, engine id) -> dict(calls, graph, static_in, feats, deferred)

    def __len__(self) -> int:
        return len(self._queue)

    # ------------------------------------------------------------------ one forward on a slot's stream
    def _eager(self, slot: int, images: torch.Tensor):
        with _slot(slot, self.chains):
            _take_deferred()
            feats = _extract(self.model, images)
            return feats, _take_deferred()

    def _forward(self, slot: int, s, images: torch.Tensor):
        """Runs on stream ``s`` (current).  Returns (features, deferred updates)."""
        if not self.graphs:
            return self._eager(slot, images)
        eng = self.model.engine() if hasattr(self.model, "engine") else None
        key = (slot, tuple(images.shape), images.dtype, bool(self.model.training), id(eng))
        ent = self._graphs.get(key)
        if ent is None:
            mine = [k for k in self._graphs if k[0] == slot]
            for k in mine[:-1] if len(mine) >= 2 else []:  # at most two shapes per slot (full batches + an epoch's ragged last one)
                del self._graphs[k]
            ent = self._graphs[key] = dict(calls=0, graph=None)
        else:
            self._graphs[key] = self._graphs.pop(key)  # most recently used last
        ent["calls"] += 1
        if ent["graph"] is None and not any(e["graph"] is not None for e in self._graphs.values()):
            # The very first forward of the pipeline: set up EVERY slot now (the set-up forwards write nothing but the slot's own buffers,
            # their deferred updates are dropped), so that no capture — each one is a device-wide sync — falls into the run later,
            # whatever the caller's warm-up length.  Later shapes (an epoch's ragged last batch) are set up lazily per slot.
            for other in range(self.depth):
                if other != slot:
                    okey = (other,) + key[1:]
                    oent = self._graphs[okey] = dict(calls=0, graph=None)
                    self._capture(other, s, images, oent, eng)
        if ent["graph"] is None:
            return self._capture(slot, s, images, ent, eng)  # (its replay computed this batch's features)
        ent["static_in"].copy_(images, non_blocking=True)
        ent["graph"].replay()
        if ent["pack"] is not None:
            ent["pack"].generation += 1  # the host code that counts rewrites of the packing does not run on a replay
        return ent["feats"], ent["deferred"]

    def _capture(self, slot: int, s, images: torch.Tensor, ent: dict, eng):
        """Set a slot up for ``images``' shape: an eager forward (allocates the slot's buffers, builds lazily cached operands — none of
        that may happen inside a capture), the capture of the very same call, and a first replay on ``images`` (a graph's first launch
        uploads it to the device: that cost belongs here, not in the run).  Returns the replay's (features, deferred updates)."""
        from .vit import lookup_pack

        self._eager(slot, images)
        static_in = torch.empty_like(images)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):  # other threads (allocator, collectives' watchdog) stay free
            feats, deferred = self._eager(slot, static_in)
        ent.update(graph=g, static_in=static_in, feats=feats, deferred=deferred,
                   pack=lookup_pack(feats) if isinstance(feats, (list, tuple)) else None,
                   # the graph holds raw addresses of the slot's buffers and of the engine's operands: both stay alive with it
                   keep=(eng, eng.slot_state(slot) if hasattr(eng, "slot_state") else None))
        static_in.copy_(images, non_blocking=True)
        g.replay()
        return feats, deferred

    def submit(self, images: torch.Tensor) -> None:
        if len(self._queue) >= self.depth:
            raise RuntimeError(f"{self.depth} forwards already in flight: call next() first")
        if self.run_ahead > 0 and len(self._issued) >= self.run_ahead:
            self._issued.popleft().synchronize()  # host waits for the forward issued ``run_ahead`` submissions ago
        if self.depth == 1:
            feats = _extract(self.model, images)
            if self.run_ahead > 0 and images.is_cuda:
                ev = torch.cuda.Event()
                ev.record()
                self._issued.append(ev)
            self._queue.append((feats, None, ()))
            return
        slot = self._n % self.depth
        s = self.streams[self._n % len(self.streams)]
        self._n += 1
        cur = torch.cuda.current_stream()
        # the batch is ready on the caller's stream, and the probe step that read this slot's buffers is already enqueued there
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            feats, deferred = self._forward(slot, s, images)
            done = torch.cuda.Event()
            done.record(s)
        if self.run_ahead > 0:
            self._issued.append(done)
        if images.is_cuda:
            images.record_stream(s)  # allocated on the caller's stream, read on the side stream
        self._queue.append((feats, done, deferred))

    def next(self):
        feats, done, deferred = self._queue.popleft()
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


def _extract(model, images):
    from .train import extract_features

    return extract_features(model, images)


def pipelined_features(model, batches: Iterable, image_key="image", depth: int = None, probe=None) -> Iterator[Tuple[object, object]]:
    """Yield ``(batch, features)`` for every batch of ``batches`` with up to ``depth`` forwards in flight: the forward of batch
    t+1 is enqueued before batch t is handed to the caller, so it runs under the caller's probe step t.  ``depth`` None:
    ``default_depth(probe)``."""
    pipe = FeaturePipeline(model, default_depth(probe) if depth is None else depth)
    if hasattr(batches, "consumer_lag"):
        # mvp.prefetch.DevicePrefetcher recycles its device buffers: this generator holds pipe.depth batches before the caller has
        # issued the probe step of the first of them (which reads that batch's target; its image is read by the forward in flight)
        batches.consumer_lag = max(int(batches.consumer_lag), pipe.depth - 1)
    it = iter(batches)
    pending = collections.deque()

    def images_of(b):
        x = b[image_key] if isinstance(b, dict) else b[0]
        if x.is_cuda or not torch.cuda.is_available():  # (a host tensor without a device: the backbone raises MvpError)
            return x
        return x.to(torch.device("cuda", torch.cuda.current_device()), non_blocking=True)

    def feed() -> bool:
        try:
            b = next(it)
        except StopIteration:
            return False
        pipe.submit(images_of(b))
        pending.append(b)
        return True

    try:
        while True:
            while len(pipe) < pipe.depth and feed():
                pass
            if not pending:
                break
            b = pending.popleft()
            yield b, pipe.next()
    finally:
        pipe.drain()
