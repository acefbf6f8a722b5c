"""Host logic of the tape-free probe step (mvp/fused_step.py) that needs no GPU: the wrapper-free LambdaLR transition against
``scheduler.step()`` (torch/optim/lr_scheduler.py), and the selection rules' refusal of what the plan does not cover."""
import torch

from evals.utils.optim import cosine_decay_linear_warmup
from mvp import fused_step


def _pair():
    def make():
        p = torch.nn.Parameter(torch.zeros(3))
        opt = torch.optim.SGD([p], lr=5e-4)
        return opt, torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda e: cosine_decay_linear_warmup(e, 200, 30))
    return make(), make()


def test_fast_lambda_lr_step_matches_scheduler_step():
    (o1, s1), (o2, s2) = _pair()
    for i in range(120):
        o1.step()
        o2.step()
        s1.step()
        fused_step.fast_scheduler_step(s2, o2)
        assert o1.param_groups[0]["lr"] == o2.param_groups[0]["lr"], i
        assert s1.get_last_lr() == s2.get_last_lr() and s1.last_epoch == s2.last_epoch and s1._step_count == s2._step_count
    d1, d2 = s1.state_dict(), s2.state_dict()
    d1.pop("lr_lambdas"), d2.pop("lr_lambdas")
    assert d1 == d2


def test_other_schedulers_go_through_their_own_step():
    p = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.SGD([p], lr=1.0)
    s = torch.optim.lr_scheduler.StepLR(opt, step_size=2, gamma=0.5)
    for _ in range(4):
        opt.step()
        fused_step.fast_scheduler_step(s, opt)
    assert opt.param_groups[0]["lr"] == 0.25 and s.last_epoch == 4


def test_plan_refuses_what_it_does_not_cover(monkeypatch):
    class Opt:  # not a FlatAdamW
        pass

    o = Opt()
    assert fused_step.plan_for(torch.nn.Identity(), o, None, torch.nn.Identity(), [torch.zeros(1)], torch.zeros(1), False) is None
    assert o._mvp_fused_plan is False  # looked at once, remembered
    assert fused_step.plan_for(None, Opt(), None, None, None, None, True) is None  # scale_invariant: the tape
    monkeypatch.setenv("MVP_FUSED_STEP", "0")
    o2 = Opt()
    assert fused_step.plan_for(None, o2, None, None, None, None, False) is None and not hasattr(o2, "_mvp_fused_plan")
