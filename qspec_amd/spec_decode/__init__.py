"""Host side of the QSpec speculative-decoding loop (mirror of vllm/spec_decode/ for this path)."""
from .engine import QSpecEngine  # noqa: F401
from .metrics import AsyncMetricsCollector, SpecDecodeWorkerMetrics  # noqa: F401
from .rejection_sampler import RejectionSampler  # noqa: F401
from .worker import (ExecuteModelRequest, SamplerOutput, SequenceGroupMetadata, SpecDecodeWorker,  # noqa: F401
                     create_spec_worker)
