"""Host logic of mvp/pipeline.py that needs no GPU: the span arithmetic (which (slot, carry) patterns consecutive span forwards cycle
through, how many images a span holds for a backbone / batch shape) and the per-model pipeline cache of ``pipelined_features``."""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "midvision-probe_amd"))

from mvp import pipeline  # noqa: E402


def test_span_patterns_cycle_through_every_cut_position_once_per_slot_parity():
    # 110 images over batches of 16: the carry goes 0, 14, 12, ... (period 8, even: every carry keeps its slot)
    assert pipeline.span_patterns(110, 16) == [(0, 0), (1, 14), (0, 12), (1, 10), (0, 8), (1, 6), (0, 4), (1, 2)]
    assert pipeline.span_patterns(104, 16) == [(0, 0), (1, 8)]
    # an odd period runs twice so that both slots see every carry
    pats = pipeline.span_patterns(7, 4)
    assert len(pats) == 4 and sorted(c for _, c in pats) == [0, 1, 2, 3]
    pats = pipeline.span_patterns(9, 6)  # carries 0, 3 (period 2)
    assert pats == [(0, 0), (1, 3)]
    for T, B in ((110, 16), (104, 64), (18, 16), (7, 4), (10, 4), (9, 6)):
        pats = pipeline.span_patterns(T, B)
        assert len(pats) % 2 == 0 and pats[0] == (0, 0) and (len(pats) * T) % B == 0
        assert all(pats[k] == (k % 2, (k * T) % B) for k in range(len(pats)))


class _Engine:
    C = 768


class _Model:
    patch_size = 16
    training = False

    def supports_grouping(self):
        return True

    def engine(self):
        return _Engine()


def test_default_span_fills_one_round_of_256_cus(monkeypatch):
    monkeypatch.delenv("MVP_PIPELINE_SPAN", raising=False)
    m = _Model()
    # ViT-B/16 at 224^2: 197 rows per image, 85 row tiles of 256 x 3 column tiles = 255 workgroups -> 110 images (a multiple of B / 8)
    assert pipeline.default_span(m, torch.empty(16, 3, 224, 224), 2, 6) == 110
    assert (110 * 197 + 255) // 256 == 85 and (111 * 197 + 255) // 256 == 86
    assert pipeline.default_span(m, torch.empty(64, 3, 224, 224), 2, 2) == 104    # multiple of 8
    assert pipeline.default_span(m, torch.empty(16, 3, 480, 640), 2, 1) == 0      # a batch fills a forward by itself: single batches
    assert pipeline.default_span(m, torch.empty(16, 3, 224, 224), 4, 6) == 0      # only the two-slot pipeline carries spans
    monkeypatch.setenv("MVP_PIPELINE_SPAN", "0")
    assert pipeline.default_span(m, torch.empty(16, 3, 224, 224), 2, 6) == 0
    monkeypatch.setenv("MVP_PIPELINE_SPAN", "96")
    assert pipeline.default_span(m, torch.empty(16, 3, 224, 224), 2, 6) == 0      # whole batches: the group covers it


def test_pipelined_features_keeps_one_pipeline_per_model_and_mode(monkeypatch):
    """The per-model cache (a weak map: nothing hangs on the module, and the entry goes with the model), on the inline path a CPU box
    can run (a model without ``supports_pipelining``: depth 1, the forward is called directly)."""
    monkeypatch.setattr(pipeline, "_extract", lambda model, images: model(images))

    class M(torch.nn.Module):
        def forward(self, x):
            return [x * 2]

    m = M()
    bs = [(torch.full((2, 3, 4, 4), float(i)),) for i in range(3)]
    out = [f for _, f in pipeline.pipelined_features(m, bs)]
    assert [float(f[0].mean()) for f in out] == [0.0, 2.0, 4.0]
    (p1,) = pipeline.cached_pipelines(m).values()
    assert not p1._lent and len(p1) == 0
    list(pipeline.pipelined_features(m, bs))
    assert list(pipeline.cached_pipelines(m).values()) == [p1]           # reused
    g = pipeline.pipelined_features(m, bs)
    next(g)                                                              # a loop suspended mid-way holds the cached pipeline ...
    list(pipeline.pipelined_features(m, bs))                             # ... so a second loop gets one of its own
    assert list(pipeline.cached_pipelines(m).values())[0] is not p1
    g.close()
    m.eval()
    list(pipeline.pipelined_features(m, bs))
    assert len(pipeline.cached_pipelines(m)) == 2                        # train / eval
    assert "_mvp_pipelines" not in m.__dict__
    n = len(pipeline._PIPELINES)
    del m, p1, g, out
    gc.collect()
    assert len(pipeline._PIPELINES) == n - 1                             # the cache entry went with the model
