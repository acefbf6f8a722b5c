"""Frozen forwards in flight on side streams (mvp/pipeline.py) must not change a single bit of the training trajectory:
the same batches through ``train_depth_step`` as one serial chain and with 2 / 3 forwards in flight give identical losses,
probe weights, AdamW state and tap-BN running statistics (those are updated in place by every forward, in batch order).
Reference semantics: train_depth.py:99-143 (one batch at a time; the backbone is frozen, so its forward of batch t+1 does not
depend on probe step t)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(dev, probe_kind="linear"):
    from evals.models.dino import DINO
    from evals.models.probes import DepthHead
    from evals.utils.optim import cosine_decay_linear_warmup
    from mvp import backbone as bb
    from mvp.optim import FlatAdamW

    model = DINO(return_multilayer=True, add_norm=True, weights=bb.random_vit_state_dict(seed=3)).to(dev)
    torch.manual_seed(11)
    if probe_kind == "linear":
        probe = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type="bindepth", min_depth=0.001, max_depth=10).to(dev)
    else:
        probe = DepthHead(feat_dim=model.feat_dim, head_type="dpt", kernel_size=3, prediction_type="bindepth", hidden_dim=128, min_depth=0.001, max_depth=10).to(dev)
    opt = FlatAdamW([{"params": probe.parameters(), "lr": 1e-3}])
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda e: cosine_decay_linear_warmup(e, 100, 10))
    return model, probe, opt, sched


def _batches(dev, n, B=4, hw=(64, 80)):
    out = []
    for s in range(n):
        g = torch.Generator().manual_seed(500 + s)
        img = torch.randn(B, 3, *hw, generator=g)
        dep = torch.rand(B, 1, *hw, generator=g) * 9.0 + 0.05
        out.append({"image": img.to(dev), "depth": dep.to(dev)})
    return out


def _run(depth, probe_kind="linear", n=7, graphs=False):
    from evals.utils.losses import DepthLoss
    from mvp.pipeline import FeaturePipeline
    from mvp.train import train_depth_step

    dev = torch.device("cuda:0")
    model, probe, opt, sched = _build(dev, probe_kind)
    loss_fn = DepthLoss()
    losses = []
    pipe = FeaturePipeline(model, depth, graphs=graphs)
    assert pipe.graphs == (graphs and depth > 1)
    bs = _batches(dev, n)
    nxt = 0
    for i in range(n):
        while len(pipe) < pipe.depth and nxt < n:
            pipe.submit(bs[nxt]["image"])
            nxt += 1
        losses.append(train_depth_step(model, probe, opt, sched, loss_fn, None, bs[i]["depth"].clone(), feats=pipe.next()))
    if graphs:  # every slot was set up (eager run + capture) at the first submit; all later forwards were replays
        assert len(pipe._graphs) == depth and all(e["graph"] is not None for e in pipe._graphs.values())
        assert sum(e["calls"] for e in pipe._graphs.values()) == n and max(e["calls"] for e in pipe._graphs.values()) >= 2
    torch.cuda.synchronize()
    bn = [torch.cat([b.running_mean, b.running_var]).cpu().numpy() for b in model.batchnorms]
    nbt = [int(b.num_batches_tracked) for b in model.batchnorms]
    return (torch.stack(losses).cpu().numpy(), opt.flat_param.cpu().numpy().copy(), opt.exp_avg_sq.cpu().numpy().copy(), bn, nbt)


@pytest.mark.parametrize("depth,graphs", [(2, False), (3, False), (2, True), (3, True), (4, True)])
def test_pipelined_training_is_bit_identical_to_serial(depth, graphs):
    """depth >= 3 also switches the backbone GEMMs to the shared-chip tiles (tile_policy): still the same bits; depth 4 runs its four
    slots on three streams (a slot's graph replays on whichever stream its turn falls on).  graphs: every slot's
    forward replays a captured hipGraph from its third call on; the tap-BN running statistics are applied by the consumer."""
    ref = _run(1)
    got = _run(depth, graphs=graphs)
    assert np.isfinite(ref[0]).all() and ref[4] == [7] * 4
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])
    np.testing.assert_array_equal(got[2], ref[2])
    for a, b in zip(got[3], ref[3]):
        np.testing.assert_array_equal(a, b)
    assert got[4] == ref[4]


def test_ragged_batches_and_graph_replay():
    """An epoch's smaller last batch changes the forward's shape: the engine then drops its buffers of the other shape, while other
    slots still hold captured graphs that address them.  The graphs keep their buffers alive; full-size batches afterwards replay the
    old graphs.  Same bits as the serial loop throughout."""
    from evals.utils.losses import DepthLoss
    from mvp.pipeline import FeaturePipeline
    from mvp.train import train_depth_step

    dev = torch.device("cuda:0")
    sizes = [4, 4, 4, 4, 2, 4, 4, 4, 2, 4, 4]

    def batches():
        out = []
        for i, b in enumerate(sizes):
            g = torch.Generator().manual_seed(900 + i)
            out.append((torch.randn(b, 3, 64, 80, generator=g).to(dev), (torch.rand(b, 1, 64, 80, generator=g) * 9 + 0.05).to(dev)))
        return out

    def run(depth, graphs):
        model, probe, opt, sched = _build(dev)
        loss_fn = DepthLoss()
        pipe = FeaturePipeline(model, depth, graphs=graphs)
        bs, losses, nxt = batches(), [], 0
        for i in range(len(bs)):
            while len(pipe) < pipe.depth and nxt < len(bs):
                pipe.submit(bs[nxt][0])
                nxt += 1
            losses.append(train_depth_step(model, probe, opt, sched, loss_fn, None, bs[i][1].clone(), feats=pipe.next()))
        torch.cuda.synchronize()
        return torch.stack(losses).cpu().numpy(), opt.flat_param.cpu().numpy().copy(), model.batchnorms[3].running_var.cpu().numpy().copy(), pipe

    ref = run(1, False)
    got = run(2, True)
    for a, b in zip(got[:3], ref[:3]):
        np.testing.assert_array_equal(a, b)
    shapes = sorted((k[0], k[1][0]) for k in got[3]._graphs)
    assert shapes == [(0, 2), (0, 4), (1, 4)], shapes  # slot 0 saw both batch sizes and keeps both graphs


def test_pipelined_dpt_probe_is_bit_identical_to_serial():
    ref = _run(1, "dpt", n=5)
    got = _run(2, "dpt", n=5, graphs=True)
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])


def test_pipeline_matches_plain_step_and_mixes_with_direct_calls():
    """pipelined_features == calling train_depth_step(images) batch by batch; a direct model(images) after a pipelined stretch
    sees the running statistics of every earlier forward (event order), and eval-mode forwards still work."""
    from evals.utils.losses import DepthLoss
    from mvp.pipeline import FeaturePipeline
    from mvp.train import train_depth_step

    dev = torch.device("cuda:0")
    bs = _batches(dev, 5)
    model, probe, opt, sched = _build(dev)
    loss_fn = DepthLoss()
    plain = [train_depth_step(model, probe, opt, sched, loss_fn, b["image"], b["depth"].clone()) for b in bs]
    ref_rm = model.batchnorms[0].running_mean.clone()
    ref_w = opt.flat_param.clone()

    model, probe, opt, sched = _build(dev)
    pipe = FeaturePipeline(model, 2, graphs=False)
    assert pipe.depth == 2
    got = []
    pipe.submit(bs[0]["image"])
    pipe.submit(bs[1]["image"])
    with pytest.raises(RuntimeError):
        pipe.submit(bs[2]["image"])
    for i in range(3):
        got.append(train_depth_step(model, probe, opt, sched, loss_fn, None, bs[i]["depth"].clone(), feats=pipe.next()))
        if i == 0:
            pipe.submit(bs[2]["image"])
    assert len(pipe) == 0
    for b in bs[3:]:  # back to the plain path on the caller's stream
        got.append(train_depth_step(model, probe, opt, sched, loss_fn, b["image"], b["depth"].clone()))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(torch.stack(got).cpu().numpy(), torch.stack(plain).cpu().numpy())
    assert torch.equal(model.batchnorms[0].running_mean, ref_rm)
    assert torch.equal(opt.flat_param, ref_w)
    model.eval()
    with torch.no_grad():
        f = model(bs[0]["image"])
    assert all(torch.isfinite(t).all() for t in f)


def test_resnet_forwards_in_flight_match_serial():
    """ResNet engines allocate per forward on the launching stream; shared state = weights + tap-BN running statistics (ordered
    by events).  Three forwards in flight give the features and running statistics of three serial calls, bit for bit."""
    from evals.models.dino_res50 import DINO_RESNET
    from mvp.pipeline import FeaturePipeline

    dev = torch.device("cuda:0")
    xs = [torch.randn(2, 3, 96, 96, generator=torch.Generator().manual_seed(40 + i)).to(dev) for i in range(3)]

    def build():
        return DINO_RESNET(return_layers=[1, 2, 3, 4], return_multilayer=True, add_norm=True, fixed_size=96, init_seed=5).to(dev)

    m = build()
    ref = [[t.clone() for t in m(x)] for x in xs]
    ref_rm = [b.running_mean.clone() for b in m.batchnorms]
    m = build()
    pipe = FeaturePipeline(m, 3, graphs=False)
    assert pipe.depth == 3 and not pipe.graphs
    for x in xs:
        pipe.submit(x)
    got = [[t.clone() for t in pipe.next()] for _ in xs]
    torch.cuda.synchronize()
    for a, b in zip(got, ref):
        for u, v in zip(a, b):
            assert torch.equal(u, v)
    for a, b in zip((bn.running_mean for bn in m.batchnorms), ref_rm):
        assert torch.equal(a, b)
    # replayed graphs (per-forward buffers live in each graph's private pool): two rounds over the same batches, the second all replays
    m = build()
    ref2 = [[t.clone() for t in m(x)] for x in xs + xs]
    ref2_rv = [b.running_var.clone() for b in m.batchnorms]
    m = build()
    pipe = FeaturePipeline(m, 2, graphs=True)
    assert pipe.graphs
    got2 = []
    for x in xs + xs:
        pipe.submit(x)
        got2.append([t.clone() for t in pipe.next()])
    torch.cuda.synchronize()
    for a, b in zip(got2, ref2):
        for u, v in zip(a, b):
            assert torch.equal(u, v)
    for a, b in zip((bn.running_var for bn in m.batchnorms), ref2_rv):
        assert torch.equal(a, b)


def test_kernels_reproduce_themselves_beside_mfma_kernels():
    """Regression for the packed-fp32 finding (csrc/Makefile): the fused bilinear x4 + bin-expectation kernel and the probe's other
    element-wise stages must return the same bits while 64x64-tile GEMMs run beside them on another stream.  With the SLP-vectorised
    build 5-10 % of these launches differed in one pixel; 300 launches each here."""
    from mvp import functional as MF, lib, ops
    from mvp.lib import PREC_BF16X3

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    a_p = ops.split_bf16(torch.randn(128, 3072, generator=g).to(dev), PREC_BF16X3)
    w_p = ops.split_bf16((torch.randn(768, 3072, generator=g) * 0.05).to(dev), PREC_BF16X3)
    out = torch.empty(128, 768, device=dev)
    side = torch.cuda.Stream()
    B, h, w, K = 4, 4, 5, 256
    l0 = torch.randn(B * h * w, K, generator=g).to(dev)
    small = (torch.rand(4, 1, 16, 20, generator=g) * 9 + 0.1).to(dev)
    x = torch.randn(84, 768, generator=g).to(dev)
    gam, bet = torch.ones(768, device=dev), torch.zeros(768, device=dev)

    def bins():
        P = B * 16 * h * w
        depth = torch.empty(B, 1, 4 * h, 4 * w, dtype=torch.float32, device=dev)
        inv = torch.empty(P, dtype=torch.float32, device=dev)
        gate = torch.empty(P, K // 8, dtype=torch.uint8, device=dev)
        lib.call("mvp_linear_bins_fwd", lib.LinearBinsArgs(lib.ptr(l0), lib.ptr(depth), lib.ptr(inv), lib.ptr(gate), None, None, B, h, w, K, 4, 0.001, 10.0))
        return [depth, inv, gate]

    def resize():
        return [MF.interpolate(small, size=(64, 80), mode="bilinear")]

    def layernorm():
        o = ops.empty_pair((84, 768), PREC_BF16X3, dev)
        ops.layernorm(x, gam, bet, o, 84, 768, 1e-6)
        return list(o)

    for fn in (bins, resize, layernorm):
        torch.cuda.synchronize()
        ref = fn()
        torch.cuda.synchronize()
        for r in range(300):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(8):
                    ops.gemm(a_p, w_p, 64, 768, 768, out_f32=out, lda=3072, ldw=3072)
            got = fn()
            assert all(torch.equal(a, b) for a, b in zip(got, ref)), f"{fn.__name__}: launch {r} differs beside the GEMM load"
        torch.cuda.synchronize()


def test_train_and_validate_loops_match_the_serial_loops(monkeypatch):
    """mvp.train.train() / validate() with their default pipelines (4 batches ahead on 3 streams, replayed graphs; eval-mode forwards
    in flight during validation) against the same loops forced serial (MVP_INFLIGHT=1): same loss history, same probe weights, same
    BN running statistics, same validation loss and metrics."""
    from evals.utils.losses import DepthLoss
    from mvp.train import train, validate

    dev = torch.device("cuda:0")

    class Loader(list):
        sampler = None

    def batches(n, seed):
        out = []
        for s in range(n):
            g = torch.Generator().manual_seed(seed + s)
            img = torch.randn(4, 3, 64, 80, generator=g)
            dep = torch.rand(4, 1, 64, 80, generator=g) * 9.0 + 0.05
            seg = torch.randint(0, 150, (4, 64, 80), generator=g)
            out.append({"image": img.to(dev), "depth": dep.to(dev), "segmentation": seg.to(dev)})
        return Loader(out)

    def run():
        model, probe, opt, sched = _build(dev)
        hist = train(model, probe, batches(9, 100), opt, sched, 2, True, DepthLoss())
        opt.finish_pending()
        model.eval(); probe.eval()
        vloss, gm, lm = validate(model, probe, batches(5, 300), DepthLoss(), verbose=False)
        torch.cuda.synchronize()
        return (hist, opt.flat_param.cpu().numpy().copy(), model.batchnorms[2].running_var.cpu().numpy().copy(), float(vloss),
                {k: float(v) for k, v in gm.items()})

    monkeypatch.setenv("MVP_INFLIGHT", "1")
    ref = run()
    monkeypatch.delenv("MVP_INFLIGHT")
    got = run()
    assert got[0] == ref[0]
    np.testing.assert_array_equal(got[1], ref[1])
    np.testing.assert_array_equal(got[2], ref[2])
    assert got[3] == ref[3]
    for k, v in ref[4].items():  # (the per-level / per-segment reductions add in hardware-atomic order: last bits may move)
        assert got[4][k] == pytest.approx(v, rel=1e-6, abs=1e-9), k


def test_prefetcher_buffers_survive_the_pipelines_look_ahead():
    """DevicePrefetcher recycles its device buffers; pipelined_features pulls 4 batches before it issues the first probe step.  The
    prefetcher must not overwrite a batch's target (read by the probe step) or image (read by the forward in flight) while the pipeline
    still holds it: 14 distinct host batches through prefetcher + pipeline == the same batches through the serial loop."""
    from evals.utils.losses import DepthLoss
    from mvp.pipeline import pipelined_features
    from mvp.prefetch import DevicePrefetcher
    from mvp.train import train_depth_step

    dev = torch.device("cuda:0")
    host = []
    for s in range(14):
        g = torch.Generator().manual_seed(700 + s)
        host.append({"image": torch.randn(4, 3, 64, 80, generator=g).pin_memory(), "depth": (torch.rand(4, 1, 64, 80, generator=g) * 9 + 0.05).pin_memory()})

    def run(depth, group):
        model, probe, opt, sched = _build(dev)
        loss_fn = DepthLoss()
        pre = DevicePrefetcher(host, dev, depth=2)
        losses = []
        for batch, feats in pipelined_features(model, pre, depth=depth, group=group):
            losses.append(train_depth_step(model, probe, opt, sched, loss_fn, None, batch["depth"], feats=feats))
        torch.cuda.synchronize()
        assert pre.consumer_lag == depth * group - 1
        return torch.stack(losses).cpu().numpy(), opt.flat_param.cpu().numpy().copy()

    ref = run(1, 1)
    for depth, group in ((4, 1), (2, 3)):  # 4 single-batch forwards ahead; 2 forwards of 3 stacked batches ahead (6 batches held)
        got = run(depth, group)
        np.testing.assert_array_equal(got[0], ref[0])
        np.testing.assert_array_equal(got[1], ref[1])


def test_leaving_the_loop_early_discards_the_forwards_in_flight():
    """A loop that breaks after 3 batches has 3 more forwards in flight: their tap-BN running-statistics updates must not be applied
    (the one-at-a-time loop would never have run them)."""
    from evals.utils.losses import DepthLoss
    from mvp.pipeline import pipelined_features
    from mvp.train import train_depth_step

    dev = torch.device("cuda:0")

    def run(depth):
        model, probe, opt, sched = _build(dev)
        loss_fn = DepthLoss()
        for i, (batch, feats) in enumerate(pipelined_features(model, _batches(dev, 9), depth=depth)):
            train_depth_step(model, probe, opt, sched, loss_fn, None, batch["depth"].clone(), feats=feats)
            if i == 2:
                break
        torch.cuda.synchronize()
        return [int(b.num_batches_tracked) for b in model.batchnorms], model.batchnorms[1].running_mean.cpu().numpy().copy(), opt.flat_param.cpu().numpy().copy()

    ref, got = run(1), run(4)
    assert ref[0] == got[0] == [3] * 4
    np.testing.assert_array_equal(got[1], ref[1])
    np.testing.assert_array_equal(got[2], ref[2])


# ---------------------------------------------------------------------------------------------------------------- grouped forwards
def _run_grouped(group, depth, n=7, graphs=False, B=4, hw=(64, 80), streams=None):
    """The same trajectory as ``_run`` with ``group`` batches stacked into every frozen forward (mvp/pipeline.py, module docstring)."""
    from evals.utils.losses import DepthLoss
    from mvp.pipeline import FeaturePipeline, pipelined_features
    from mvp.train import train_depth_step

    dev = torch.device("cuda:0")
    model, probe, opt, sched = _build(dev)
    loss_fn = DepthLoss()
    bs = _batches(dev, n, B=B, hw=hw)
    pipe = FeaturePipeline(model, depth, graphs=graphs, group=group, streams=streams)
    losses = [train_depth_step(model, probe, opt, sched, loss_fn, None, b["depth"].clone(), feats=f) for b, f in pipelined_features(model, bs, pipe=pipe)]
    torch.cuda.synchronize()
    bn = [torch.cat([b.running_mean, b.running_var]).cpu().numpy() for b in model.batchnorms]
    nbt = [int(b.num_batches_tracked) for b in model.batchnorms]
    return (torch.stack(losses).cpu().numpy(), opt.flat_param.cpu().numpy().copy(), opt.exp_avg_sq.cpu().numpy().copy(), bn, nbt), pipe


@pytest.mark.parametrize("group,depth,graphs", [(3, 2, False), (2, 2, True), (3, 2, True), (4, 3, True)])
def test_grouped_forwards_are_bit_identical_to_serial(group, depth, graphs):
    """7 batches through groups of 3 + 3 + 1 (2 + 2 + 2 + 1, 4 + 3): the batches of a group share every launch of the frozen forward
    except the tap BN (train-mode statistics of ONE batch, dino.py:185-191), whose running-statistics updates are applied per batch
    in batch order.  Losses, probe weights, AdamW state, running statistics and step counters equal the one-batch-at-a-time loop's
    (train_depth.py:99-143) bit for bit; with graphs, full groups replay a captured hipGraph and the ragged last group runs eagerly."""
    ref = _run(1)
    got, pipe = _run_grouped(group, depth, graphs=graphs)
    assert pipe.group == group and pipe.depth == depth
    if graphs:
        assert len(pipe._graphs) == depth and all(e["graph"] is not None and k[-1] == group for k, e in pipe._graphs.items())
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])
    np.testing.assert_array_equal(got[2], ref[2])
    for a, b in zip(got[3], ref[3]):
        np.testing.assert_array_equal(a, b)
    assert got[4] == ref[4] == [7] * 4


@pytest.mark.parametrize("span,graphs,n", [(10, False, 9), (10, True, 9), (6, True, 8), (14, True, 12), (9, True, 13), (7, True, 11)])
def test_span_forwards_are_bit_identical_to_serial(span, graphs, n):
    """Forwards over SPANS of the image stream that end in the middle of a batch (mvp/pipeline.py, "Spans"): batches of 4 images,
    ``span`` images per forward, so most forwards start with the rest of the batch the previous one cut (after 2 images for the even
    spans; 9 and 7 cycle through all four cut positions).  The tap BN of that batch still runs over its own 4 images (the engine carries the cut batch's tap-level rows into the
    next forward), so losses, probe weights, AdamW state, running statistics and step counters equal the one-batch-at-a-time loop's
    (train_depth.py:99-143) bit for bit — through two legs of the same pipeline (the second leg restarts on a batch boundary), with
    graph replay of the full spans and eager short ones."""
    from evals.utils.losses import DepthLoss
    from mvp.pipeline import FeaturePipeline, pipelined_features
    from mvp.train import train_depth_step

    dev = torch.device("cuda:0")
    ref = _run(1, n=n)
    model, probe, opt, sched = _build(dev)
    loss_fn = DepthLoss()
    bs = _batches(dev, n)
    pipe = FeaturePipeline(model, 2, graphs=graphs, group=-(-span // 4), span=span)
    losses = []
    for part in (bs[:3], bs[3:]):  # a warm-up leg shorter than two spans, then the rest
        for b, f in pipelined_features(model, part, pipe=pipe):
            losses.append(train_depth_step(model, probe, opt, sched, loss_fn, None, b["depth"].clone(), feats=f))
    assert pipe.span == span and pipe.depth == 2 and pipe.chains == 1
    if graphs:  # one graph per (slot, carry) pattern, all captured at the first submit
        from mvp.pipeline import span_patterns
        assert all(e["graph"] is not None for e in pipe._graphs.values())
        assert sorted((k[0], k[-1].carry) for k in pipe._graphs) == sorted(span_patterns(span, 4))
        assert sum(e["calls"] for e in pipe._graphs.values()) >= 1
    torch.cuda.synchronize()
    np.testing.assert_array_equal(torch.stack(losses).cpu().numpy(), ref[0])
    np.testing.assert_array_equal(opt.flat_param.cpu().numpy(), ref[1])
    np.testing.assert_array_equal(opt.exp_avg_sq.cpu().numpy(), ref[2])
    for bnm, r in zip(model.batchnorms, ref[3]):
        np.testing.assert_array_equal(torch.cat([bnm.running_mean, bnm.running_var]).cpu().numpy(), r)
    assert [int(b.num_batches_tracked) for b in model.batchnorms] == [n] * 4


def test_span_forwards_with_a_ragged_last_batch():
    """An epoch whose last batch is smaller (drop_last=False): the span that would run into it ends on the batch boundary instead, and
    the ragged batch runs as a forward of its own — five batches of 4 and one of 3, spans of 6 images: forwards of 6, 6, 6, 2 and 3
    images.  Trajectory = the one-batch-at-a-time loop's, bit for bit."""
    from evals.utils.losses import DepthLoss
    from mvp.pipeline import FeaturePipeline, pipelined_features
    from mvp.train import train_depth_step

    dev = torch.device("cuda:0")

    def run(span):
        model, probe, opt, sched = _build(dev)
        bs = _batches(dev, 6)
        bs[-1] = {k: v[:3].contiguous() for k, v in bs[-1].items()}
        pipe = FeaturePipeline(model, 2 if span else 1, graphs=bool(span), group=2 if span else 1, span=span or None)
        losses = [train_depth_step(model, probe, opt, sched, DepthLoss(), None, b["depth"].clone(), feats=f)
                  for b, f in pipelined_features(model, bs, pipe=pipe)]
        torch.cuda.synchronize()
        return (torch.stack(losses).cpu().numpy(), opt.flat_param.cpu().numpy().copy(),
                [torch.cat([m.running_mean, m.running_var]).cpu().numpy() for m in model.batchnorms])

    ref, got = run(0), run(6)
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])
    for a, b in zip(got[2], ref[2]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("output", ["dense", "dense-cls"])
def test_span_forwards_in_eval_mode_return_each_batchs_own_features(output):
    """validate()'s shape: the backbone in eval mode (tap BN from running statistics, no deferred updates), spans of 7 images over
    batches of 4: every batch's features equal ``model(images)`` of that batch alone, bit for bit, 'dense-cls' (the tap kernel's CLS
    output next to the map) included."""
    from evals.models.dino import DINO
    from mvp import backbone as bb
    from mvp.pipeline import FeaturePipeline, pipelined_features

    dev = torch.device("cuda:0")
    model = DINO(return_multilayer=True, add_norm=True, output=output, weights=bb.random_vit_state_dict(seed=3)).to(dev)
    for i, bnm in enumerate(model.batchnorms):  # non-trivial running statistics
        g = torch.Generator().manual_seed(40 + i)
        bnm.running_mean.copy_(torch.randn(bnm.running_mean.shape, generator=g).to(dev) * 0.1)
        bnm.running_var.copy_((torch.rand(bnm.running_var.shape, generator=g) + 0.5).to(dev))
    model.eval()
    bs = _batches(dev, 6)
    ref = [[t.clone() for t in model(b["image"])] for b in bs]
    pipe = FeaturePipeline(model, 2, graphs=True, group=2, span=7)
    got = [[t.clone() for t in f] for _, f in pipelined_features(model, bs, pipe=pipe)]
    assert pipe.span == 7 and len(got) == 6
    for a, b in zip(got, ref):
        for x, y in zip(a, b):
            assert torch.equal(x, y)


def test_two_span_pipelines_over_one_backbone_keep_their_carries_apart():
    """A loop suspended in the middle of a cut batch (its first images' tap rows wait in the engine's carry store) while ANOTHER pipeline
    over the same backbone runs span forwards of its own — a validation pass inside a training epoch: each image stream has its own
    carry store (pipeline.Span.stream), so both get every batch's own features."""
    from evals.models.dino import DINO
    from mvp import backbone as bb
    from mvp.pipeline import FeaturePipeline, pipelined_features

    dev = torch.device("cuda:0")
    model = DINO(return_multilayer=True, add_norm=True, weights=bb.random_vit_state_dict(seed=3)).to(dev).eval()
    a, b = _batches(dev, 6), [{k: v.flip(0) * 0.5 for k, v in x.items()} for x in _batches(dev, 4)]
    ref_a = [[t.clone() for t in model(x["image"])] for x in a]
    ref_b = [[t.clone() for t in model(x["image"])] for x in b]
    ga = pipelined_features(model, a, pipe=FeaturePipeline(model, 2, graphs=True, group=2, span=7))
    got_a = [[t.clone() for t in next(ga)[1]] for _ in range(2)]  # spans of 7 + 7 images are in flight: batches 1 and 3 are cut
    got_b = [[t.clone() for t in f] for _, f in pipelined_features(model, b, pipe=FeaturePipeline(model, 2, graphs=True, group=2, span=7))]
    got_a += [[t.clone() for t in f] for _, f in ga]
    assert len(got_a) == 6 and len(got_b) == 4
    for got, ref in ((got_a, ref_a), (got_b, ref_b)):
        for x, y in zip(got, ref):
            for u, v in zip(x, y):
                assert torch.equal(u, v)


def test_epochs_reuse_the_models_cached_pipeline_and_rebind_to_new_shapes():
    """``pipelined_features(model, loader)`` without an explicit pipeline keeps ONE pipeline per model and mode (train / eval): the second
    epoch replays the graphs the first one captured instead of capturing again, a validation pass in eval mode gets its own, and a loader
    with another image size re-shapes the cached pipeline (``rebind``).  Every batch's features equal the plain forward's."""
    from evals.models.dino import DINO
    from mvp import backbone as bb
    from mvp.pipeline import cached_pipelines, pipelined_features

    dev = torch.device("cuda:0")
    model = DINO(return_multilayer=True, add_norm=True, weights=bb.random_vit_state_dict(seed=3)).to(dev).eval()
    bs = _batches(dev, 20, B=4, hw=(32, 48))  # 7 tokens per image: spans of 32 images would be whole batches -> groups of 8
    ref = [[t.clone() for t in model(b["image"])] for b in bs]

    def epoch(batches, refs):
        got = [[t.clone() for t in f] for _, f in pipelined_features(model, batches)]
        assert len(got) == len(refs)
        for a, b in zip(got, refs):
            for x, y in zip(a, b):
                assert torch.equal(x, y)

    epoch(bs, ref)
    (pipe,) = cached_pipelines(model).values()
    assert pipe.depth == 2 and pipe.group > 1 and pipe.graphs
    ngraphs, calls = len(pipe._graphs), sum(e["calls"] for e in pipe._graphs.values())
    epoch(bs, ref)
    assert list(cached_pipelines(model).values()) == [pipe] and len(pipe._graphs) == ngraphs
    assert sum(e["calls"] for e in pipe._graphs.values()) > calls  # replays, no new capture
    big = _batches(dev, 5, B=4, hw=(64, 80))
    epoch(big, [[t.clone() for t in model(b["image"])] for b in big])
    assert list(cached_pipelines(model).values()) == [pipe] and pipe._resolved_for[0] == (4, 3, 64, 80)
    model.train()
    n_train = sum(1 for _ in pipelined_features(model, bs[:3]))
    import copy
    assert n_train == 3 and len(cached_pipelines(model)) == 2
    assert len(cached_pipelines(copy.deepcopy(model))) == 0  # nothing of the pipelines hangs on the module itself


def test_replayed_forwards_keep_their_token_packings_registered_across_other_pipelines(monkeypatch):
    """The probe head finds the token-major packing of a feature list through ``mvp.vit.lookup_pack``; a graph replay does not run the host
    code that registers it.  With the registry squeezed to what ONE pipeline needs (2 slots x 8 batches per forward), a validation pass
    (its own pipeline, its own packings) between two training epochs ages the training pipeline's entries out — the replay must bring them back (``FeaturePipeline._forward``
    re-registers its packings), else the head would silently re-pack the NCHW maps on every step from then on."""
    from evals.models.dino import DINO
    from mvp import backbone as bb
    from mvp import vit
    from mvp.pipeline import pipelined_features

    dev = torch.device("cuda:0")
    model = DINO(return_multilayer=True, add_norm=True, weights=bb.random_vit_state_dict(seed=3)).to(dev)
    bs = _batches(dev, 16, B=4, hw=(32, 48))
    monkeypatch.setattr(vit, "_PACK_REGISTRY_MAX", 16)

    def epoch():
        hits = 0
        for _, f in pipelined_features(model, bs):
            hits += int(vit.lookup_pack(f) is not None)
        return hits

    model.train()
    assert epoch() == len(bs)
    model.eval()
    assert epoch() == len(bs)  # the eval pipeline's packings push the training pipeline's out of the 16-entry registry
    model.train()
    assert epoch() == len(bs)  # replays only: every packing is found again


def test_default_span_of_the_timed_configuration():
    """B = 16 at 224^2 on ViT-B/16: 110 images per forward (21670 rows = 85 x 3 tiles of 256^2: one round of 256 CUs for the
    N = 768 GEMMs); 480x640 (1201 rows per image): single batches on three streams, as before (measured faster than 18-image spans)."""
    from mvp.pipeline import FeaturePipeline

    dev = torch.device("cuda:0")
    model, _, _, _ = _build(dev)
    pipe = FeaturePipeline(model, None, group=None)
    pipe.resolve_group(torch.empty(16, 3, 224, 224, device=dev))
    assert (pipe.group, pipe.depth, pipe.chains, pipe.span) == (7, 2, 1, 110)
    pipe = FeaturePipeline(model, None, group=None)
    pipe.resolve_group(torch.empty(16, 3, 480, 640, device=dev))
    assert (pipe.group, pipe.depth, pipe.chains, pipe.span) == (1, 4, 3, 0)


def test_warmup_shorter_than_a_group_sets_the_full_group_graphs_up():
    """bench.py's shape: a warm-up of fewer batches than one group, then full groups.  The full-group graphs of every slot are captured
    at the pipeline's FIRST submit (on copies of that batch, updates dropped), so the later full groups only replay."""
    from evals.utils.losses import DepthLoss
    from mvp.pipeline import FeaturePipeline, pipelined_features
    from mvp.train import train_depth_step

    dev = torch.device("cuda:0")
    ref = _run(1, n=8)
    model, probe, opt, sched = _build(dev)
    loss_fn = DepthLoss()
    bs = _batches(dev, 8)
    pipe = FeaturePipeline(model, 2, graphs=True, group=3)
    losses = []
    for leg, part in enumerate((bs[:2], bs[2:])):  # warm-up of 2 (one ragged group), then 3 + 3
        for b, f in pipelined_features(model, part, pipe=pipe):
            losses.append(train_depth_step(model, probe, opt, sched, loss_fn, None, b["depth"].clone(), feats=f))
        if leg == 0:
            assert len(pipe._graphs) == 2 and all(e["graph"] is not None for e in pipe._graphs.values())
            calls = sum(e["calls"] for e in pipe._graphs.values())
    assert sum(e["calls"] for e in pipe._graphs.values()) == calls + 2  # the two full groups replayed; nothing new was captured
    torch.cuda.synchronize()
    np.testing.assert_array_equal(torch.stack(losses).cpu().numpy(), ref[0])
    np.testing.assert_array_equal(opt.flat_param.cpu().numpy(), ref[1])
    assert [int(b.num_batches_tracked) for b in model.batchnorms] == [8] * 4


def test_timed_configuration_b16_224_grouped_graphs_vs_serial_and_oracle():
    """What bench.py times by default (VERDICT r2 #2b): B = 16, 224^2, six batches per frozen forward (M = 18912 token rows: the
    large-M ping-pong GEMM kernel, csrc/gemm_pp.hip), hipGraph replay, 2 forwards in flight.  (a) the trajectory over 8 batches
    (groups 6 + 2) equals the serial one-batch-at-a-time loop bit for bit; (b) the tap features of a batch INSIDE a group equal the
    CPU oracle's (reference restatement, oracle/vit.py) to <= 1e-3 rel-L2, and its loss the oracle trainer's."""
    from evals.models.dino import DINO
    from evals.models.probes import DepthHead
    from evals.utils.losses import DepthLoss
    from mvp.optim import FlatAdamW
    from mvp.pipeline import FeaturePipeline, pipelined_features
    from mvp.train import train_depth_step
    from oracle import probes as oprobes
    from oracle import train as otrain
    from oracle import vit as ovit

    dev = torch.device("cuda:0")
    vsd = ovit.make_vit_weights(seed=0)
    psd = oprobes.make_linear_head_weights([768] * 4, 256, 1, seed=3)
    B, n = 16, 8
    host = [otrain.synthetic_depth_batch(B, 224, 224, rank=0, step=s) for s in range(n)]
    bs = [(i.to(dev), t.to(dev)) for i, t in host]

    def run(depth, group, graphs, span=0, keep=4):
        model = DINO(return_multilayer=True, add_norm=True, weights=vsd).to(dev)
        probe = DepthHead(feat_dim=model.feat_dim, head_type="linear", kernel_size=1, prediction_type="bindepth")
        probe.load_state_dict(psd, strict=True)
        probe = probe.to(dev)
        opt = FlatAdamW([{"params": probe.parameters(), "lr": 5e-4}])
        pipe = FeaturePipeline(model, depth, graphs=graphs, group=group, span=span)
        pipe.resolve_group(bs[0][0])
        assert pipe.span == span
        losses, kept = [], None
        for i, ((img, tgt), f) in enumerate(pipelined_features(model, bs, pipe=pipe)):
            if i == keep:
                kept = [t.clone() for t in f]  # features of the fifth batch: inside the first group of six
            losses.append(train_depth_step(model, probe, opt, None, DepthLoss(), None, tgt.clone(), feats=f))
        torch.cuda.synchronize()
        return torch.stack(losses).cpu().numpy(), opt.flat_param.cpu().numpy().copy(), [k.cpu().numpy() for k in kept], [b.running_var.cpu().numpy() for b in model.batchnorms]

    ref = run(1, 1, False)
    got = run(2, 6, True)
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])
    for a, b in zip(got[2], ref[2]):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(got[3], ref[3]):
        np.testing.assert_array_equal(a, b)
    # (c) the default of round 3's end: spans of 110 images (M = 21670 rows; forwards of 110 + 18 images here), the kept batch (the
    # seventh) is the one the first span cuts (after 14 images)
    ref6 = run(1, 1, False, keep=6)
    sp = run(2, 7, True, span=110, keep=6)
    np.testing.assert_array_equal(sp[0], ref[0])
    np.testing.assert_array_equal(sp[1], ref[1])
    for a, b in zip(sp[2], ref6[2]):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(sp[3], ref[3]):
        np.testing.assert_array_equal(a, b)
    # (b) against the CPU oracle: batch 4 alone (its tap BN statistics are its own), first-step loss of the trajectory
    tr = otrain.DepthProbeTrainer(vsd, psd, max_step=100, warmup_step=10)
    feats_ref = tr.features(host[4][0])
    for f, fr in zip(got[2], feats_ref):
        d = f.astype(np.float64) - fr.numpy().astype(np.float64)
        assert np.sqrt((d * d).sum() / (fr.numpy().astype(np.float64) ** 2).sum()) < 1e-3
    loss0, _ = tr.forward_loss(tr.features(host[0][0]), host[0][1].clone())
    assert abs(got[0][0] - loss0.item()) < 2e-3 * abs(loss0.item())


def test_streamk_workspace_is_per_slot_under_graph_replay(monkeypatch):
    """With MVP_STREAMK=1 every plain backbone GEMM takes the stream-K kernel, whose partial-tile workspace must not be shared by
    forwards that replay side by side.  The workspace is keyed by pipeline slot (not by the stream a slot was captured on: every slot's
    hipGraph is captured on one stream and replayed on rotating ones — ADVICE r2): pipelined + replayed == serial, bit for bit."""
    from mvp import ops

    monkeypatch.setattr(ops, "_STREAMK_MODE", "1")
    monkeypatch.setattr(ops, "_STREAMK_WS", {})
    ref = _run(1)
    got = _run(4, graphs=True)
    assert len({k for k in ops._STREAMK_WS if len(k) == 3 and k[1] == "slot"}) == 4  # one workspace per slot
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])
