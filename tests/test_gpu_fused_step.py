"""The tape-free probe step (mvp/fused_step.py) against the autograd path of ``train_depth_step`` (train_depth.py:99-143): the same
batches give the same losses, probe weights, AdamW moments, .grad aliases and LR-scheduler state bit for bit — it issues the same
launches — serial and with forwards in flight; probes / options it does not cover stay on the tape."""
import numpy as np
import pytest
import torch

from test_gpu_pipeline import _batches, _build

pytestmark = pytest.mark.gpu


def _run(monkeypatch, fused, depth=1, n=7, probe_kind="linear", hook=False, scale_invariant=False):
    from evals.utils.losses import DepthLoss
    from mvp import fused_step
    from mvp.pipeline import FeaturePipeline
    from mvp.train import train_depth_step

    monkeypatch.setenv("MVP_FUSED_STEP", "1" if fused else "0")
    dev = torch.device("cuda:0")
    model, probe, opt, sched = _build(dev, probe_kind)
    seen = []
    if hook:
        probe.register_forward_hook(lambda m, i, o: seen.append(1))
    loss_fn = DepthLoss()
    pipe = FeaturePipeline(model, depth, graphs=depth > 1)
    bs, losses, nxt = _batches(dev, n), [], 0
    for i in range(n):
        while len(pipe) < pipe.depth and nxt < n:
            pipe.submit(bs[nxt]["image"])
            nxt += 1
        losses.append(train_depth_step(model, probe, opt, sched, loss_fn, None, bs[i]["depth"].clone(), feats=pipe.next(), scale_invariant=scale_invariant))
    torch.cuda.synchronize()
    plan = getattr(opt, "_mvp_fused_plan", None)
    w = probe.head.conv.weight if probe_kind == "linear" else None
    alias = None if w is None else (w.grad.data_ptr() == opt.flat_grad.data_ptr() or w.grad.data_ptr() == opt.grad_slots()[0][1].data_ptr())
    state = dict(losses=torch.stack(losses).cpu().numpy(), param=opt.flat_param.cpu().numpy().copy(), m=opt.exp_avg.cpu().numpy().copy(),
                 v=opt.exp_avg_sq.cpu().numpy().copy(), grad=opt.flat_grad.cpu().numpy().copy(), lr=opt.param_groups[0]["lr"],
                 sched=dict((k, v) for k, v in sched.state_dict().items() if k != "lr_lambdas"), steps=opt._step)
    return state, isinstance(plan, fused_step.LinearBinsDepthStep), alias, len(seen)


def _same(a, b):
    for k in ("losses", "param", "m", "v", "grad"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert a["lr"] == b["lr"] and a["sched"] == b["sched"] and a["steps"] == b["steps"]


@pytest.mark.parametrize("depth", [1, 2])
def test_tape_free_step_is_bit_identical_to_the_autograd_step(monkeypatch, depth):
    ref, used_ref, alias_ref, _ = _run(monkeypatch, False, depth)
    got, used, alias, _ = _run(monkeypatch, True, depth)
    assert not used_ref and used, "MVP_FUSED_STEP=0 must take the tape; the default must take the plan for the linear bindepth probe"
    assert alias_ref and alias  # weight.grad aliases its slot of the flat gradient after either step
    assert np.isfinite(ref["losses"]).all() and ref["steps"] == 7
    _same(got, ref)


def test_other_probes_hooks_and_scale_invariant_stay_on_the_tape(monkeypatch):
    _, used, _, _ = _run(monkeypatch, True, probe_kind="dpt", n=2)
    assert not used
    ref, _, _, _ = _run(monkeypatch, False, n=3)
    got, used, _, calls = _run(monkeypatch, True, n=3, hook=True)
    assert not used and calls == 3  # a registered forward hook must keep firing: nn.Module.__call__ is on the tape path only
    _same(got, ref)
    _, used, _, _ = _run(monkeypatch, True, n=2, scale_invariant=True)
    assert not used


def test_bn_running_update_n_equals_single_launches():
    """mvp_bn_running_update_n (all tap BNs of a batch in one launch) against one mvp_bn_running_update per module: same bits, step
    counters included; a module listed twice is refused by the C entry point and serialised by the Python wrapper."""
    from mvp import lib, ops

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    Cs = [768, 768, 300, 768, 1024]

    def state():
        gg = torch.Generator().manual_seed(6)
        return [(torch.randn(C, generator=gg).to(dev), (torch.rand(C, generator=gg) + 0.5).to(dev), torch.tensor(3 + i, dtype=torch.int64, device=dev)) for i, C in enumerate(Cs)]

    stats = [torch.randn(3 * C, generator=g).abs().to(dev) for C in Cs]
    a, b = state(), state()
    for (rm, rv, nbt), st, C in zip(a, stats, Cs):
        ops.bn_running_update(st, rm, rv, nbt, C)
    ops.bn_running_update_many([(st, rm, rv, nbt if i != 2 else None, C) for i, ((rm, rv, nbt), st, C) in enumerate(zip(b, stats, Cs))])
    torch.cuda.synchronize()
    for i, ((rm, rv, nbt), (rm2, rv2, nbt2)) in enumerate(zip(a, b)):
        assert torch.equal(rm, rm2) and torch.equal(rv, rv2)
        assert int(nbt2) == (int(nbt) if i != 2 else int(nbt) - 1)
    rm, rv, nbt = b[0]
    arr = (lib.BnRunningUpdateArgs * 2)(*[lib.BnRunningUpdateArgs(lib.ptr(stats[0]), lib.ptr(rm), lib.ptr(rv), None, Cs[0], 0.1)] * 2)
    assert lib.load().mvp_bn_running_update_n(arr, 2, lib.stream_ptr()) != 0  # MVP_EINVAL: the same module twice
    assert lib.load().mvp_bn_running_update_n(arr, 9, lib.stream_ptr()) != 0
    before = rm.clone()
    ops.bn_running_update_many([(stats[0], rm, rv, None, Cs[0])] * 2)  # wrapper: two ordered launches
    ref = before.clone()
    for _ in range(2):
        r2, v2 = ref.clone(), rv.clone()
        ops.bn_running_update(stats[0], r2, v2, None, Cs[0])
        ref = r2
    assert torch.equal(rm, ref)
