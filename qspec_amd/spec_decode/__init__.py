"""Host side of the QSpec speculative-decoding loop (mirror of vllm/spec_decode/ for this path)."""
from .engine import QSpecEngine  # noqa: F401
from .metrics import AsyncMetricsCollector, SpecDecodeWorkerMetrics  # noqa: F401
from .rejection_sampler import RejectionSampler  # noqa: F401
from .typical_acceptance_sampler import TypicalAcceptanceSampler  # noqa: F401
from .worker import (CompletionSequenceGroupOutput, DeviceHandoffTimeout, ExecuteModelRequest, Logprob,  # noqa: F401
                     SamplerOutput, SamplingParams, SequenceData, SequenceGroupMetadata, SequenceOutput,
                     SpecDecodeWorker, SpeculativeConfig, create_spec_worker)
