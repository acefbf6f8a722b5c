"""Host -> HBM input pipeline.  The reference moves each batch with a blocking `.to(rank)` inside the
step (train_depth.py:102-104) and syncs on `.item()` every iteration; at 3 ms per step the 13 MB pageable copy alone
would be a sixth of the step.  DevicePrefetcher keeps `depth` batches in flight: batch k+1 is copied into a reused
device buffer on a side HIP stream while step k computes (true async DMA when the loader pins its batches); `next()` makes the compute stream wait on the
copy's event only (no host sync).  Device buffers are recycled once the consumer asks for the batch after next (later for a
look-ahead consumer: ``consumer_lag``)."""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, Optional

import torch


class DevicePrefetcher:
    def __init__(self, loader: Iterable[Dict[str, object]], device: Optional[torch.device] = None, depth: int = 2, keys=None,
                 consumer_lag: int = 0):
        """``consumer_lag``: how many batches the consumer may hold BEYOND the one it is working on before it has issued their work on
        its stream.  0 = the plain loop (when batch k+1 is asked for, the step of batch k has been enqueued).  A look-ahead consumer —
        mvp.pipeline.pipelined_features pulls ``depth`` batches before it issues the first probe step — sets it to ``depth - 1``
        (it does so itself through this attribute): device buffers are then recycled that many pulls later."""
        if not torch.cuda.is_available():
            raise RuntimeError("DevicePrefetcher needs a HIP device (no CPU fallback on the product path)")
        self.loader, self.depth, self.keys = loader, max(1, int(depth)), keys
        self.consumer_lag = max(0, int(consumer_lag))
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.stream = torch.cuda.Stream(device=self.device)
        self._dev, self._h2d_done = {}, {}

    def __len__(self):
        return len(self.loader)

    @property
    def sampler(self):  # train() calls loader.sampler.set_epoch(...)
        return getattr(self.loader, "sampler", None)

    def _stage(self, slot: int, batch: Dict[str, object]):
        out = {}
        done = self._h2d_done.get(slot)
        if done is not None:
            done.synchronize()  # host: the previous H2D of this slot has finished (normally long ago)
        with torch.cuda.stream(self.stream):
            for k, v in batch.items():
                if not torch.is_tensor(v) or (self.keys is not None and k not in self.keys):
                    out[k] = v
                    continue
                key = (slot, k)
                dst = self._dev.get(key)
                if dst is None or dst.shape != v.shape or dst.dtype != v.dtype:
                    dst = self._dev[key] = torch.empty(v.shape, dtype=v.dtype, device=self.device)
                # pinned source (DataLoader(pin_memory=True), builder.py:58): true async DMA on the side stream.
                # pageable source: the runtime stages it itself (host-blocking but ~6x faster than an explicit
                # pageable -> pinned torch copy on this thread, measured 4663 vs 802 img/s at B=16).
                dst.copy_(v, non_blocking=True)
                out[k] = dst
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._h2d_done[slot] = ev
        return out, ev

    def __iter__(self) -> Iterator[Dict[str, object]]:
        it = iter(self.loader)
        lag = self.consumer_lag
        nslot = self.depth + 1 + lag  # the batches the consumer holds (1 + lag) + `depth` in flight
        ready_ev = [None] * nslot  # compute-stream events marking "slot no longer read"
        queue = []
        k = 0

        def fill():
            nonlocal k
            try:
                batch = next(it)
            except StopIteration:
                return False
            slot = k % nslot
            if ready_ev[slot] is not None:
                self.stream.wait_event(ready_ev[slot])  # do not overwrite device buffers a running step still reads
            queue.append((slot,) + self._stage(slot, batch))
            k += 1
            return True

        for _ in range(self.depth):
            if not fill():
                break
        handed = []  # slots in the order their batches were handed to the consumer
        while queue:
            slot, out, ev = queue.pop(0)
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for v in out.values():  # allocated on the side stream, consumed on `cur`: tell the caching allocator
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(cur)
            if len(handed) > lag:  # the step of the batch handed over 1 + lag pulls ago has been enqueued on `cur` by now
                e = torch.cuda.Event()
                e.record(cur)
                ready_ev[handed[-1 - lag]] = e
            fill()
            handed.append(slot)
            if len(handed) > nslot:
                handed.pop(0)
            yield out
