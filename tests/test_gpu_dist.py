"""N>1 on the PRODUCT path (train_depth.py:64-73, 620-622, 851-855): two ranks run
DINO -> DepthHead -> DepthLoss -> FlatAdamW.step() with the flat-gradient all-reduce, sharing cuda:0 over gloo
(a one-GPU box cannot host two RCCL ranks; the collective call site, the broadcast and the overlap logic are the same code).

Asserted: (1) replicas stay bit-identical although they were seeded differently (rank-0 broadcast) and see different
shards; (2) they equal the single-process emulation: per-shard forward/backward (per-shard tap-BN statistics and DepthLoss,
SURVEY §8e), SUM of the flat gradients, one AdamW with grad_scale 1/2; (3) overlap_comm on/off give the same bits."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(out_dir, overlap, world=2):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MVP_DIST_BACKEND="gloo", MVP_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_rank.py"), str(out_dir), str(int(overlap))], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(os.path.join(out_dir, f"rank{r}.npz")) for r in range(world)]


def _emulate(world=2):
    import _dist_rank as R
    from evals.utils.losses import DepthLoss
    from mvp import functional as MF

    dev = torch.device("cuda:0")
    model, probe, opt, sched = R.build(dev, probe_seed=100, overlap=False)  # rank 0's seed: what the broadcast distributes
    loss_fn = DepthLoss()
    losses = [[] for _ in range(world)]
    for s in range(R.STEPS):
        grads = []
        for r in range(world):
            images, tgt = R.batch(r, s, dev)
            opt.zero_grad()
            with torch.no_grad():
                feats = [f.detach() for f in model(images)]
            pred = MF.interpolate(probe(feats), size=tgt.shape[-2:], mode="bilinear")
            loss = loss_fn(pred, tgt)
            loss.backward()
            opt._gather_stray_grads()
            grads.append(opt.flat_grad.clone())
            losses[r].append(loss.item())
        opt.flat_grad.copy_(grads[0] + grads[1])  # what a 2-rank SUM all-reduce delivers (fp32 addition commutes)
        opt._step += 1
        opt._apply(opt.flat_grad, world, float(opt.param_groups[0]["lr"]), opt._step)
        sched.step()
    torch.cuda.synchronize()
    return opt.flat_param.cpu().numpy(), np.array(losses)


@pytest.mark.timeout(900)
def test_two_rank_product_step_matches_mean_gradient_emulation(tmp_path):
    assert torch.cuda.is_available()
    ref_flat, ref_losses = _emulate()
    for overlap in (False, True):
        d = tmp_path / f"ov{int(overlap)}"
        d.mkdir()
        r0, r1 = _run_ranks(d, overlap)
        assert int(r0["world"]) == 2 and str(r0["backend"]) == "gloo"
        np.testing.assert_array_equal(r0["flat"], r1["flat"])  # replicas bit-identical (incl. the rank-0 broadcast)
        np.testing.assert_array_equal(r0["flat"], ref_flat)    # == mean-gradient emulation, bit for bit
        np.testing.assert_array_equal(r0["losses"], ref_losses[0])
        np.testing.assert_array_equal(r1["losses"], ref_losses[1])
        assert not np.array_equal(r0["losses"], r1["losses"])  # the ranks really saw different shards


def test_rccl_world1_overlapped_allreduce_with_graph_replay(tmp_path):
    """The RCCL branch of FlatAdamW on the one GPU this pool has: init_process_group("nccl", world_size=1) and the
    ``force_comm`` hook send every step's flat gradient through a real RCCL all_reduce(async_op=True) -> work.wait() -> AdamW, with the
    pipeline's grouped forwards replaying captured hipGraphs beside RCCL's stream and watchdog thread.  The trajectory (losses, weights,
    AdamW second moments) is bit-identical to the non-distributed one.  (Reference: DDP wrap of the probe, train_depth.py:620-622.)"""
    out = tmp_path / "w1.npz"
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_world1.py"), str(out)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=420)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
    r = np.load(out)
    assert str(r["backend"]) == "nccl"
    assert np.isfinite(r["la"]).all()
    np.testing.assert_array_equal(r["lb"], r["la"])
    np.testing.assert_array_equal(r["pb"], r["pa"])
    np.testing.assert_array_equal(r["vb"], r["va"])
