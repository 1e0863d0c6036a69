"""Acceptance metrics (qspec_amd/spec_decode/metrics.py).  The scenarios are those of the reference's
tests/spec_decode/test_metrics.py (first call None, second call after the interval returns metrics, non-zero ranks
never collect, nothing before the interval elapses, the published ratios), against this build's collector whose
counters are one device tensor [accepted, emitted, draft] written by the rejection kernels."""
import math
from types import SimpleNamespace
from unittest.mock import MagicMock

import pytest
import torch


def test_ratio_formulas():
    from qspec_amd.spec_decode.metrics import get_max_num_emitted_tokens, metrics_from_counters
    m = metrics_from_counters(3931, 5161, 4092, 3)          # the numbers of the reference's own screenshot
    assert round(m.draft_acceptance_rate, 3) == 0.961 and round(m.system_efficiency, 3) == 0.946
    assert get_max_num_emitted_tokens(4092, 3) == 5456
    z = metrics_from_counters(0, 0, 0, 5)
    assert math.isnan(z.draft_acceptance_rate) and math.isnan(z.system_efficiency)
    m = metrics_from_counters(10, 15, 20, 5)                 # test_initial_metrics_has_correct_values' case
    assert m.accepted_tokens == 10 and m.emitted_tokens == 15 and m.draft_tokens == 20 and m.num_spec_tokens == 5
    assert m.draft_acceptance_rate == 10 / 20 and m.system_efficiency == 15 / 24


def _collector(counters, timer_values, interval=5.0, rank=0):
    from qspec_amd.spec_decode.metrics import AsyncMetricsCollector
    timer = MagicMock()
    timer.side_effect = list(timer_values)
    c = AsyncMetricsCollector(SimpleNamespace(counters=counters), timer=timer, collect_interval_s=interval)
    c.init_gpu_tensors(rank=rank)
    return c


@pytest.mark.gpu
def test_first_call_none_second_call_metrics():
    counters = torch.tensor([10, 15, 20], dtype=torch.long, device="cuda:0")
    c = _collector(counters, [0.0, 5.1, 5.2])
    assert c.maybe_collect_rejsample_metrics(k=5) is None        # schedules the copy
    m = c.maybe_collect_rejsample_metrics(k=5)
    assert m is not None and (m.accepted_tokens, m.emitted_tokens, m.draft_tokens) == (10, 15, 20)
    assert m.draft_acceptance_rate == 0.5 and m.system_efficiency == 15 / 24


@pytest.mark.gpu
@pytest.mark.parametrize("rank", [1, 2, 7])
def test_nonzero_rank_never_collects(rank):
    counters = torch.zeros(3, dtype=torch.long, device="cuda:0")
    c = _collector(counters, [0.0, 5.1, 5.2, 10.3], rank=rank)
    assert c.maybe_collect_rejsample_metrics(k=5) is None
    assert c.maybe_collect_rejsample_metrics(k=5) is None


@pytest.mark.gpu
def test_nothing_before_the_interval_elapses():
    counters = torch.zeros(3, dtype=torch.long, device="cuda:0")
    c = _collector(counters, [0.0, 4.9, 4.95, 5.2, 5.3])
    assert c.maybe_collect_rejsample_metrics(k=5) is None     # 4.9 < 5.0: no copy scheduled
    assert c.maybe_collect_rejsample_metrics(k=5) is None     # 4.95
    assert c.maybe_collect_rejsample_metrics(k=5) is None     # 5.2: copy scheduled
    assert c.maybe_collect_rejsample_metrics(k=5) is not None
