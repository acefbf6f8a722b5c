"""Host logic of mvp/pipeline.py that needs no device: batch/feature pairing, bounded look-ahead, inline mode for backbones
without per-slot buffers, draining when the consumer stops early."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "midvision-probe_amd"))


class _Recorder(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.calls = []

    def forward(self, x):
        self.calls.append(int(x[0]))
        return [x * 2.0, x * 3.0]


def _stream(n, log):
    for i in range(n):
        log.append(("load", i))
        yield {"image": torch.full((2,), float(i)), "depth": torch.full((2,), float(-i))}


def test_inline_pipeline_pairs_batches_with_their_features():
    from mvp.pipeline import FeaturePipeline, pipelined_features

    m = _Recorder()
    assert FeaturePipeline(m, 3).depth == 1  # no supports_pipelining attribute -> inline on the caller's stream
    log = []
    seen = []
    for batch, feats in pipelined_features(m, _stream(5, log), depth=3):
        assert isinstance(feats, list) and not feats[0].requires_grad
        assert torch.equal(feats[0], batch["image"] * 2.0) and torch.equal(feats[1], batch["image"] * 3.0)
        seen.append(int(batch["image"][0]))
        log.append(("step", seen[-1]))
    assert seen == [0, 1, 2, 3, 4] and m.calls == seen
    # depth 1: a batch is loaded and run only when the previous step has been issued
    assert log == [x for i in range(5) for x in (("load", i), ("step", i))]


def test_pipeline_bounds_and_early_exit():
    from mvp.pipeline import FeaturePipeline, pipelined_features

    m = _Recorder()
    pipe = FeaturePipeline(m, 1)
    pipe.submit(torch.zeros(2))
    with pytest.raises(RuntimeError):
        pipe.submit(torch.zeros(2))
    pipe.next()
    assert len(pipe) == 0
    with pytest.raises(ValueError):
        FeaturePipeline(m, 0)
    gen = pipelined_features(m, _stream(4, []), depth=1)
    next(gen)
    gen.close()  # consumer stops early: nothing left queued, no exception
    assert pipelined_features  # (generator finalised above)


def test_default_depth_env(monkeypatch):
    from mvp import pipeline

    monkeypatch.delenv("MVP_INFLIGHT", raising=False)
    assert pipeline.default_depth() == 4
    monkeypatch.setenv("MVP_INFLIGHT", "1")
    assert pipeline.default_depth() == 1
    monkeypatch.setenv("MVP_INFLIGHT", "0")
    assert pipeline.default_depth() == 1


def test_default_depth_by_probe(monkeypatch):
    from mvp import pipeline

    class P:
        def __init__(self, name):
            self.name = name

    monkeypatch.delenv("MVP_INFLIGHT", raising=False)
    assert pipeline.default_depth(P("bindepth_linear_k1")) == 4
    assert pipeline.default_depth(P("bindepth_dpt_k3")) == 1
    assert pipeline.default_depth(P("snorm_dpt_k3_UA")) == 1
    monkeypatch.setenv("MVP_INFLIGHT", "3")
    assert pipeline.default_depth(P("bindepth_dpt_k3")) == 3
