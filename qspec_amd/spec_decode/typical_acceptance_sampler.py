"""TypicalAcceptanceSampler with the reference's call surface
(vllm/model_executor/layers/typical_acceptance_sampler.py:8-172; base class SpecDecodeDeterministicBaseSampler,
spec_decode_base_sampler.py:9-131,257), backed by two HIP launches (csrc/sampler.hip).  Selected by
`draft_token_acceptance_method = "typical_acceptance_sampler"` (vllm/spec_decode/spec_decode_worker.py:95-110).

Deterministic (no random draws): a draft token is accepted iff the target's probability of it exceeds
min(posterior_threshold, posterior_alpha * exp(-entropy of the target distribution)); the replacement at the first
rejected position is the target's argmax.
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops
from .rejection_sampler import RejectionSampler


class TypicalAcceptanceSampler(RejectionSampler):
    """(Inherits the counters / device-tensor bookkeeping of the shared base class; `forward` is its own.)"""

    def __init__(self, posterior_threshold: float, posterior_alpha: float, strict_mode: bool = False, seed: int = 0):
        super().__init__(strict_mode=strict_mode, seed=seed)
        self._posterior_threshold = float(posterior_threshold)
        self._posterior_alpha = float(posterior_alpha)

    def forward(self, target_with_bonus_probs: torch.Tensor, bonus_token_ids: torch.Tensor,
                draft_probs: Optional[torch.Tensor], draft_token_ids: torch.Tensor, seeded_seqs=None, *,
                out: Optional[torch.Tensor] = None, accepted: Optional[torch.Tensor] = None,
                recovered: Optional[torch.Tensor] = None, uniform=None, exponential=None,
                active_lens: Optional[torch.Tensor] = None) -> torch.Tensor:
        """-> output_token_ids [B, k+1] (int64; -1 = no token).  draft_probs is unused (typical_acceptance_sampler.py:62);
        seeded_seqs / uniform / exponential are accepted for call compatibility with the rejection sampler and ignored."""
        B, k1, V = target_with_bonus_probs.shape
        k = k1 - 1
        dev = target_with_bonus_probs.device
        if self._strict_mode:
            assert draft_token_ids.shape == (B, k) and bonus_token_ids.numel() == B
            assert target_with_bonus_probs.dtype == torch.float32 and draft_token_ids.dtype == torch.int64
        if B == 0:
            return torch.empty(0, k + 1, device=dev, dtype=torch.int64)
        out = out if out is not None else torch.empty(B, k + 1, dtype=torch.int64, device=dev)
        accepted = accepted if accepted is not None else torch.empty(B, k, dtype=torch.uint8, device=dev)
        recovered = recovered if recovered is not None else torch.empty(B, k, dtype=torch.int64, device=dev)
        bonus = bonus_token_ids.squeeze(-1) if bonus_token_ids.dim() == 2 else bonus_token_ids
        ops.typical_acceptance_sample(target_with_bonus_probs, bonus, draft_token_ids, self._posterior_threshold,
                                      self._posterior_alpha, out, accepted, recovered, self.counters, active_lens=active_lens)
        return out

    __call__ = forward
