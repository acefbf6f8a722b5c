"""N>1 path on CPU (gloo, world_size 2): sharding semantics and the single flat-gradient
all-reduce.  The compute inside each rank is the CPU oracle (the HIP path needs a GPU); what is
tested here is the product's distributed host logic (mvp.dist) and the data-parallel contract:

    W-way sharded step  ==  single-process emulation over the same W micro-batches with
    per-shard tap-BN statistics, per-shard DepthLoss (quirk Q1 is a per-shard batch quantity)
    and averaged gradients                                        (SURVEY §4 / §8e).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import rel_l2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_indices_match_distributed_sampler():
    from torch.utils.data import DistributedSampler
    from mvp.dist import shard_indices

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 37

        def __getitem__(self, i):
            return i

    for world in (1, 2, 4, 8):
        for epoch in (0, 3):
            for rank in range(world):
                samp = DistributedSampler(DS(), num_replicas=world, rank=rank, shuffle=True, seed=0)
                samp.set_epoch(epoch)
                assert list(samp) == shard_indices(37, rank, world, epoch=epoch, shuffle=True, seed=0)
                samp = DistributedSampler(DS(), num_replicas=world, rank=rank, shuffle=False)
                assert list(samp) == shard_indices(37, rank, world, shuffle=False)


def _make_trainer():
    from oracle import probes as oprobes
    from oracle import train as otrain
    from oracle import vit as ovit

    D = 128
    vsd = ovit.make_vit_weights(embed_dim=D, depth=2, seed=41)
    psd = oprobes.make_linear_head_weights([D] * 2, 256, 1, seed=42)
    return otrain.DepthProbeTrainer(vsd, psd, layers=(0, 1), heads=2, max_step=20, warmup_step=2)


def _rank_main(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    from mvp import dist as mdist
    from oracle import train as otrain

    mdist.ddp_setup(rank, world, port, backend="gloo")
    tr = _make_trainer()
    shapes = [tr.probe_sd[n].shape for n in tr.names]
    offs, total = mdist.flat_layout(shapes)

    def hook(grads):
        flat = torch.zeros(total)
        for (o, n), g in zip(offs, grads):
            flat[o:o + n] = g.reshape(-1)
        w = mdist.all_reduce_sum_flat(flat)
        assert w == world
        return [(flat[o:o + n] / w).reshape(g.shape) for (o, n), g in zip(offs, grads)]

    losses = []
    for step in range(2):
        images, tgt = otrain.synthetic_depth_batch(3, 64, 64, rank=rank, step=step)
        losses.append(tr.step(images, tgt, grad_hook=hook))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), w=tr.probe_sd["head.conv.weight"].detach().numpy(),
             b=tr.probe_sd["head.conv.bias"].detach().numpy(), losses=np.array(losses))
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process_emulation(tmp_path):
    from oracle import train as otrain

    world = 2
    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["w"], r1["w"])  # replicas stay bit-identical
    np.testing.assert_array_equal(r0["b"], r1["b"])

    # single-process emulation: two trainers sharing one set of probe weights
    trs = [_make_trainer() for _ in range(world)]
    for step in range(2):
        grads = []
        for r, tr in enumerate(trs):
            images, tgt = otrain.synthetic_depth_batch(3, 64, 64, rank=r, step=step)
            for p in tr.probe_sd.values():
                p.grad = None
            loss, _ = tr.forward_loss(tr.features(images), tgt)
            loss.backward()
            grads.append([tr.probe_sd[n].grad.clone() for n in tr.names])
        mean = [sum(g[i] for g in grads) / world for i in range(len(trs[0].names))]
        for tr in trs:
            tr.step_from_grads(mean) if hasattr(tr, "step_from_grads") else None
        from oracle import optim as ooptim
        for tr in trs:
            lr = tr.lr_at(tr.t)
            tr.t += 1
            with torch.no_grad():
                ooptim.adamw_step([tr.probe_sd[n] for n in tr.names], mean, tr.m, tr.v, tr.t, lr)
    assert rel_l2(r0["w"], trs[0].probe_sd["head.conv.weight"].detach().numpy()) < 1e-6
    assert rel_l2(r0["b"], trs[0].probe_sd["head.conv.bias"].detach().numpy()) < 1e-6
