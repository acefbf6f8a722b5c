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


def _run_ranks(out_dir, overlap, world=2, mode=None):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MVP_DIST_BACKEND="gloo", MVP_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_rank.py"), str(out_dir), str(int(overlap))] + ([mode] if mode else []), env=env,
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


def _emulate(world=2, steps=None):
    import _dist_rank as R
    from evals.utils.losses import DepthLoss
    from mvp import functional as MF

    dev = torch.device("cuda:0")
    model, probe, opt, sched = R.build(dev, probe_seed=100, overlap=False)  # rank 0's seed: what the broadcast distributes
    loss_fn = DepthLoss()
    losses = [[] for _ in range(world)]
    for s in range(R.STEPS if steps is None else steps):
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


@pytest.mark.timeout(900)
def test_two_rank_span_pipelines_with_graph_replay_match_mean_gradient_emulation(tmp_path):
    """The branch an 8-GPU job takes when hipGraph replay is switched on (MVP_PIPELINE_GRAPHS=1; multi-rank jobs default to eager
    launches): every rank keeps its OWN span pipeline — forwards of 7 images over batches of 3 on a side stream, the (slot, carry)
    graphs captured at the first submit and replayed, a batch carried across forwards — while step t's flat-gradient all-reduce
    (async, overlapped) is still pending.  Two ranks share cuda:0 over gloo (slow: their queues are time-sliced; a correctness run).
    Replicas stay bit-identical and equal the single-process mean-gradient emulation on the serial loop, bit for bit.
    (Reference: train_depth.py:620-622, DDP wrap of the probe; 99-144, the loop.)"""
    import _dist_rank as R

    assert torch.cuda.is_available()
    ref_flat, ref_losses = _emulate(steps=R.PIPE_STEPS)
    r0, r1 = _run_ranks(tmp_path, True, mode="pipe")
    assert int(r0["world"]) == 2 and str(r0["backend"]) == "gloo"
    np.testing.assert_array_equal(r0["flat"], r1["flat"])
    np.testing.assert_array_equal(r0["flat"], ref_flat)
    np.testing.assert_array_equal(r0["losses"], ref_losses[0])
    np.testing.assert_array_equal(r1["losses"], ref_losses[1])


def test_spair_pair_sharding_two_ranks_equals_single_process(tmp_path):
    """BASELINE config #5 shards image PAIRS over the ranks (evaluate_spair_correspondence.py:104-121 is the single loop):
    two ranks (gloo, one shared card) each evaluate pairs r, r + 2, ... through their own forward pipeline, gather the per-pair error /
    index vectors with one all_gather_object and re-sort them into dataset order — every rank must return exactly what one process
    returns for the whole dataset (recall and confusion matrix, bit for bit; 19 pairs: uneven shards), with its forwards replaying captured
    hipGraphs (no collective is in flight inside that loop, so it keeps graph replay at any world size)."""
    import _spair_rank as S
    from mvp import spair

    assert torch.cuda.is_available()
    recall, conf = spair.evaluate_dataset(S.build(torch.device("cuda:0")), S.dataset(), 0.10)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MVP_DIST_BACKEND="gloo", MVP_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2",
                   MVP_INFLIGHT="2", MVP_PIPELINE_GROUP="2")  # two forwards (of two pairs) in flight per rank: the sharded loop keeps hipGraph replay at world > 1 (mvp/spair.py)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_spair_rank.py"), str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, o.decode(errors="replace")[-3000:]
    for r in range(2):
        got = np.load(os.path.join(tmp_path, f"spair{r}.npz"))
        assert int(got["world"]) == 2 and str(got["backend"]) == "gloo"
        assert float(got["recall"]) == recall
        np.testing.assert_array_equal(got["conf"], conf.numpy())
        assert got["graphs"].all() and (got["depth"] == 2).all() and int(got["replays"].sum()) >= 1, (got["graphs"], got["depth"], got["replays"])
    assert conf.sum().item() > 0 and 0.0 <= recall <= 100.0


@pytest.mark.parametrize("mode", ["groups", "spans"])
def test_rccl_world1_overlapped_allreduce_with_graph_replay(tmp_path, mode):
    """The RCCL branch of FlatAdamW on the one GPU this pool has: init_process_group("nccl", world_size=1) and the
    ``force_comm`` hook send every step's flat gradient through a real RCCL all_reduce(async_op=True) -> work.wait() -> AdamW, with the
    pipeline's grouped forwards replaying captured hipGraphs beside RCCL's stream and watchdog thread.  The trajectory (losses, weights,
    AdamW second moments) is bit-identical to the non-distributed one.  (Reference: DDP wrap of the probe, train_depth.py:620-622.)"""
    out = tmp_path / "w1.npz"
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    # mode "spans": the pipeline shape multi-GPU jobs run when graphs are switched on — span forwards (7 images over batches of 3), the
    # pre-captured (slot, carry) graphs, a batch carried across forwards, and a ragged last batch whose eager forwards run beside the
    # all-reduce still pending from the previous step
    p = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_world1.py"), str(out), mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=420)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
    r = np.load(out)
    assert str(r["backend"]) == "nccl"
    assert np.isfinite(r["la"]).all()
    np.testing.assert_array_equal(r["lb"], r["la"])
    np.testing.assert_array_equal(r["pb"], r["pa"])
    np.testing.assert_array_equal(r["vb"], r["va"])
