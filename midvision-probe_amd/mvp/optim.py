"""Flat fused AdamW for the probe parameters (train_depth.py:624-627 uses torch.optim.AdamW
defaults: betas (0.9, 0.999), eps 1e-8, weight_decay 0.01).

All parameters are re-pointed into ONE contiguous fp32 buffer, their .grad into a second
one: the optimiser step is a single HIP kernel over the flat buffer and the data-parallel
gradient exchange is a single RCCL all-reduce of the flat gradient.
It subclasses torch.optim.Optimizer so torch LR schedulers (LambdaLR, train_depth.py:636-641)
drive ``param_groups[0]["lr"]`` exactly as they do for torch.optim.AdamW.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch

from . import ops


# parameter id -> (flat gradient buffer, element offset, numel, shape): lets the probe-head backward kernels write a gradient
# straight into its slot of the flat buffer (no zero-fill before, no accumulate-add / copy after: see FlatAdamW.zero_grad)
_GRAD_DST = {}


def grad_destination(param: torch.Tensor):
    """A FRESH view of ``param``'s slot in its optimiser's flat gradient buffer, or None.  A backward that returns this very tensor
    lets autograd adopt it as ``param.grad`` (it steals a gradient it holds the only reference to), so the gradient never moves."""
    ent = _GRAD_DST.get(id(param))
    if ent is None:
        return None
    flat, off, n, shape, ref = ent
    if ref() is not param:  # id reuse after the parameter died
        return None
    if param.grad is not None:
        # a second backward() before step() (gradient accumulation): the slot already holds g1 and param.grad aliases it — writing g2
        # there and letting autograd add "the returned gradient" to param.grad would yield 2 * g2.  A fresh tensor makes autograd
        # accumulate g1 + g2 (FlatAdamW.zero_grad drops .grad, so the trainers' one-backward-per-step path still writes in place).
        return None
    return flat[off:off + n].view(shape)


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, process_group=None,
                 overlap_comm: bool = False, broadcast_init: bool = True, force_comm: bool = False):
        """``overlap_comm``: start the flat-gradient all-reduce asynchronously in ``step()`` (on a copy of the gradient, so
        ``zero_grad`` may run at once) and apply AdamW in ``finish_pending()`` — the trainers call that right before the next
        probe forward, so the collective runs under the next step's frozen backbone forward (which does not read the probe
        weights).  The reference overlaps the same exchange with backward through DDP buckets (train_depth.py:620-622).
        ``broadcast_init``: rank 0's parameters are broadcast at construction, as the DDP wrap does.
        ``force_comm`` (test hook, also MVP_FORCE_COMM=1): take the overlapped collective path at world size 1 too — a one-GPU box can
        then run the real RCCL all-reduce, its stream hand-over (``work.wait()``) and the deferred AdamW (tests/test_gpu_dist.py)."""
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise NotImplementedError("FlatAdamW handles the single probe parameter group of the trainers (model_lr == 0)")
        plist = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        if not plist:
            raise ValueError("no trainable parameters")
        dev = plist[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdamW needs device parameters (no CPU fallback)")
        sizes = [(p.numel() + 3) // 4 * 4 for p in plist]  # keep every view 16-byte aligned
        total = sum(sizes)
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p, sz in zip(plist, sizes):
                n = p.numel()
                self.flat_param[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[off:off + n].view(p.shape)
                p.grad = self.flat_grad[off:off + n].view(p.shape)
                off += sz
        import weakref

        off = 0
        for p, sz in zip(plist, sizes):
            _GRAD_DST[id(p)] = (self.flat_grad, off, p.numel(), tuple(p.shape), weakref.ref(p))
            weakref.finalize(p, _GRAD_DST.pop, id(p), None)  # the table must not keep a dead optimiser's flat buffer alive
            off += sz
        self._params = plist
        self._n = total
        self._step = 0
        self.process_group = process_group
        self.overlap_comm = bool(overlap_comm)
        import os

        self.force_comm = bool(force_comm) or os.environ.get("MVP_FORCE_COMM") == "1"
        self._pending = None  # (work handle | None, world, lr, step) of a started-but-unapplied update
        self._comm_buf = None
        from .dist import world_size as _world

        if broadcast_init and _world(process_group) > 1:
            import torch.distributed as dist

            dist.broadcast(self.flat_param, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0, group=process_group)

    def zero_grad(self, set_to_none: bool = True):
        """Drop the .grad references instead of zero-filling the flat buffer: every probe parameter receives a full gradient each
        step, which the head kernels write straight into the flat buffer (grad_destination) or which step() copies there; a
        parameter that received none has its slot zeroed in step().  No fill kernel, no accumulate-add kernels."""
        for p in self._params:
            p.grad = None

    def _gather_stray_grads(self):
        """Bring every gradient into its slot of the flat buffer: already there (written in place by the kernels) -> nothing;
        elsewhere (generic autograd path, e.g. the DPT head) -> one copy; absent -> zero the slot."""
        off = 0
        for p in self._params:
            n = p.numel()
            want = self.flat_grad.data_ptr() + off * 4
            if p.grad is None:
                self.flat_grad[off:off + n].zero_()
                p.grad = self.flat_grad[off:off + n].view(p.shape)
            elif p.grad.data_ptr() != want:
                self.flat_grad[off:off + n].copy_(p.grad.reshape(-1))
                p.grad = self.flat_grad[off:off + n].view(p.shape)
            off += (n + 3) // 4 * 4

    def all_reduce_grads(self):
        """One collective per step: SUM over ranks of the flat fp32 gradient (RCCL over xGMI);
        the 1/world average is folded into the AdamW kernel's grad_scale."""
        from .dist import all_reduce_sum_flat

        return all_reduce_sum_flat(self.flat_grad, self.process_group)

    def _apply(self, grad: torch.Tensor, world: int, lr: float, step: int):
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        # schedule state goes BY VALUE with the launch: a pinned staging buffer would be overwritten by a host that
        # runs several steps ahead of the stream before the earlier async copy has read it
        ops.adamw_step(self.flat_param, grad, self.exp_avg, self.exp_avg_sq, None, self._n,
                       beta1=b1, beta2=b2, eps=g["eps"], weight_decay=g["weight_decay"], grad_scale=1.0 / world,
                       lr=lr, bias_c1=1.0 - b1 ** step, bias_c2=1.0 - b2 ** step)

    @torch.no_grad()
    def finish_pending(self):
        """Wait (stream-side) for the in-flight gradient all-reduce and apply the AdamW update it belongs to.
        No-op when nothing is pending.  Called by the trainers before the probe forward, by ``state_dict`` and by ``step``."""
        if self._pending is None:
            return
        work, world, lr, step = self._pending
        self._pending = None
        if work is not None:
            work.wait()  # nccl: the current stream waits for the collective; gloo: host wait
        self._apply(self._comm_buf, world, lr, step)

    @torch.no_grad()
    def step(self, closure=None):
        self.finish_pending()
        self._gather_stray_grads()
        return self._step_flat()

    def step_in_place(self):
        """``step()`` for a caller whose kernels have just written EVERY gradient into its slot of the flat buffer (the tape-free
        probe step, mvp/fused_step.py): no walk over the parameters, and none of torch.optim.Optimizer's per-call wrappers (profiler
        range, hook dispatch, the LR scheduler's call tracker) — that caller has checked that no step hook is registered.  Call under
        ``torch.no_grad()``."""
        self.finish_pending()
        self._opt_called = True  # what the LR scheduler's wrapper of step() records (its first-call ordering warning reads it)
        return self._step_flat()

    def _step_flat(self):
        self._step += 1
        lr = float(self.param_groups[0]["lr"])
        from .dist import world_size as _world

        world = _world(self.process_group)
        if self.overlap_comm and (world > 1 or (self.force_comm and torch.distributed.is_available() and torch.distributed.is_initialized())):
            import torch.distributed as dist

            if self._comm_buf is None:
                self._comm_buf = torch.empty_like(self.flat_grad)
            self._comm_buf.copy_(self.flat_grad)
            work = dist.all_reduce(self._comm_buf, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)
            self._pending = (work, world, lr, self._step)
            return None
        world = self.all_reduce_grads()
        self._apply(self.flat_grad, world, lr, self._step)
        return None

    def grad_slots(self):
        """[(parameter, its slot of the flat gradient as a view of the parameter's shape)], in flat-buffer order."""
        out, off = [], 0
        for p in self._params:
            n = p.numel()
            out.append((p, self.flat_grad[off:off + n].view(p.shape)))
            off += (n + 3) // 4 * 4
        return out

    def state_dict(self):
        self.finish_pending()
        return super().state_dict()
