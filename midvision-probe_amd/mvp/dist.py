"""One process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm) for rendezvous and
the single per-step collective.  Mirrors train_depth.py:64-73 (ddp_setup) and the
DistributedSampler sharding of evals/datasets/builder.py:50-65."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def ddp_setup(rank: int, world_size: int, port: int, backend: str = "nccl"):
    """Reference: train_depth.py:64-73.  127.0.0.1 instead of 'localhost' (container hostnames may not resolve)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend=backend, rank=rank, world_size=world_size)
    if backend == "nccl":
        torch.cuda.set_device(rank % max(torch.cuda.device_count(), 1))


def env_setup(backend: str = "nccl"):
    """torchrun-style: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment."""
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", rank))
    # Rehearsal overrides (1-GPU boxes): MVP_DIST_BACKEND=gloo + MVP_FORCE_DEVICE=0 let several ranks share
    # one card to exercise the multi-rank control flow; production = nccl (RCCL), one rank per GPU.
    backend = os.environ.get("MVP_DIST_BACKEND", backend)
    dev = int(os.environ.get("MVP_FORCE_DEVICE", local))
    if torch.cuda.is_available():
        torch.cuda.set_device(dev)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def all_reduce_sum_flat(flat: torch.Tensor, group=None) -> int:
    """The ONE data-path collective of a training step: SUM over ranks of the flat fp32 probe
    gradient (RCCL over xGMI on GPUs, gloo in the CPU tests).  Returns the world size; the caller
    folds 1/world into the optimiser kernel.  No-op (returns 1) when not distributed."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return dist.get_world_size(group)
    return 1


def flat_layout(shapes):
    """Offsets of each parameter inside the flat buffer (every view 16-byte aligned)."""
    offs, off = [], 0
    for shp in shapes:
        n = 1
        for d in shp:
            n *= int(d)
        offs.append((off, n))
        off += (n + 3) // 4 * 4
    return offs, off


def shard_indices(n: int, rank: int, world: int, epoch: int = 0, shuffle: bool = True, seed: int = 0):
    """DistributedSampler semantics (builder.py:50-51, train_depth.py:94-95): permutation seeded
    by seed+epoch, padded by wrap-around to a multiple of world, rank r takes r, r+W, ..."""
    if shuffle:
        g = torch.Generator().manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    total = (n + world - 1) // world * world
    pad = total - len(idx)
    if pad > 0:
        idx += (idx * ((pad + len(idx) - 1) // len(idx)))[:pad]
    return idx[rank:total:world]
