"""vllm/spec_decode/metrics.py: SpecDecodeWorkerMetrics (:12-45) and AsyncMetricsCollector (:51-209).

The counters live on the GPU (the rejection kernels add to them); the collector copies them on a side stream
every `collect_interval_s` and turns them into the two published ratios:
    draft_acceptance_rate = accepted / draft            (:171-174; `accepted` counted non-causally,
                                                          vllm/model_executor/layers/spec_decode_base_sampler.py:127)
    system_efficiency     = emitted / ((draft / k) * (k + 1))                               (:176-188,190-209)
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import Callable, Optional

import torch


@dataclass
class SpecDecodeWorkerMetrics:
    draft_acceptance_rate: float
    system_efficiency: float
    draft_tokens: int
    emitted_tokens: int
    accepted_tokens: int
    num_spec_tokens: int


Timer = Callable[[], float]


def get_max_num_emitted_tokens(draft_tokens: int, k: int) -> int:
    assert draft_tokens % k == 0
    return (draft_tokens // k) * (k + 1)


def metrics_from_counters(accepted: int, emitted: int, draft: int, k: int) -> SpecDecodeWorkerMetrics:
    max_emitted = get_max_num_emitted_tokens(draft, k)
    return SpecDecodeWorkerMetrics(
        draft_acceptance_rate=accepted / draft if draft > 0 else float("nan"),
        system_efficiency=emitted / max_emitted if max_emitted > 0 else float("nan"),
        draft_tokens=draft, emitted_tokens=emitted, accepted_tokens=accepted, num_spec_tokens=k)


class AsyncMetricsCollector:
    def __init__(self, spec_decode_sampler, timer: Optional[Timer] = None, collect_interval_s: float = 5.0):
        self.spec_decode_sampler = spec_decode_sampler
        self._timer = time.time if timer is None else timer
        self._rank: Optional[int] = None
        self._copy_stream: Optional[torch.cuda.Stream] = None
        self._in_flight_copy: Optional[torch.cuda.Event] = None
        self._host = torch.zeros(3, dtype=torch.long, device="cpu").pin_memory() if torch.cuda.is_available() \
            else torch.zeros(3, dtype=torch.long)
        self._rejsample_metrics_collect_interval_s = collect_interval_s
        self._last_metrics_collect_time = self._timer()

    def init_gpu_tensors(self, rank: int) -> None:
        self._rank = rank
        # off CUDA-alike platforms the reference skips the collection altogether (vllm/spec_decode/metrics.py:96-98)
        self._copy_stream = torch.cuda.Stream() if torch.cuda.is_available() else None

    def maybe_collect_rejsample_metrics(self, k: int) -> Optional[SpecDecodeWorkerMetrics]:
        if self._in_flight_copy is not None:
            ready, self._in_flight_copy = self._in_flight_copy, None
            return self._collect_rejsample_metrics(k, ready)
        if self._should_collect_rejsample_metrics(self._timer()):
            self._in_flight_copy = self._copy_rejsample_metrics_async()
        return None

    def _should_collect_rejsample_metrics(self, now: float) -> bool:
        if self._rank != 0 or self._copy_stream is None:
            return False
        return now - self._last_metrics_collect_time >= self._rejsample_metrics_collect_interval_s

    def _copy_rejsample_metrics_async(self) -> torch.cuda.Event:
        assert self._copy_stream is not None
        self._copy_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._copy_stream):
            self._host.copy_(self.spec_decode_sampler.counters, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(self._copy_stream)
        return ev

    def _collect_rejsample_metrics(self, k: int, ready_event: torch.cuda.Event) -> SpecDecodeWorkerMetrics:
        ready_event.synchronize()
        self._last_metrics_collect_time = self._timer()
        accepted, emitted, draft = (int(v) for v in self._host.tolist())
        return metrics_from_counters(accepted, emitted, draft, k)
