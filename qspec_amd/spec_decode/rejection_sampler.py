"""RejectionSampler with the reference's call surface
(vllm/model_executor/layers/rejection_sampler.py:27-154, base class spec_decode_base_sampler.py:9-131),
backed by two HIP kernels instead of ~15 torch ops over [B,k,V].

Random draws: the reference uses torch.rand / exponential_ on the CUDA Philox stream, which cannot be
reproduced off NVIDIA; here they come from an in-kernel Philox4x32-10 keyed by (seed, offset) held on the
device, or are injected (tests) -- see include/qspec_hip.h.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import ops


class RejectionSampler:
    def __init__(self, strict_mode: bool = False, use_flashinfer: Optional[bool] = None, seed: int = 0):
        self._strict_mode = strict_mode
        self._num_bonus_tokens = 1
        self.counters: Optional[torch.Tensor] = None     # [accepted, emitted, draft] on the device
        self.rng_state: Optional[torch.Tensor] = None    # [seed, offset] on the device
        self._seed = seed

    # names of spec_decode_base_sampler.py:33-58
    def init_gpu_tensors(self, device) -> None:
        assert self.counters is None
        if isinstance(device, int):
            device = f"cuda:{device}"
        self.counters = torch.zeros(3, dtype=torch.long, device=device)
        self.rng_state = torch.tensor([self._seed, 0], dtype=torch.long, device=device)

    init_tensors = init_gpu_tensors

    @property
    def num_accepted_tokens(self):
        return self.counters[0]

    @property
    def num_emitted_tokens(self):
        return self.counters[1]

    @property
    def num_draft_tokens(self) -> int:
        return int(self.counters[2].item())

    @property
    def probs_dtype(self):
        return torch.float32

    @property
    def token_id_dtype(self):
        return torch.int64

    def forward(self, target_with_bonus_probs: torch.Tensor, bonus_token_ids: torch.Tensor,
                draft_probs: torch.Tensor, draft_token_ids: torch.Tensor,
                seeded_seqs: Optional[Dict[int, torch.Generator]] = None, *, out: Optional[torch.Tensor] = None,
                accepted: Optional[torch.Tensor] = None, recovered: Optional[torch.Tensor] = None,
                uniform: Optional[torch.Tensor] = None, exponential: Optional[torch.Tensor] = None,
                active_lens: Optional[torch.Tensor] = None) -> torch.Tensor:
        """-> output_token_ids [B, k+1] (int64; -1 = no token).
        seeded_seqs {row: torch.Generator} (rejection_sampler.py:60-67,229-250,387-399): those rows draw their
        uniforms and exponentials from their own generator, the others from the default stream -- done with torch on
        the slow (eager) path and handed to the kernels as injected draws.
        active_lens [B] int32 (engine): rows with active_lens <= 0 are empty batch slots -- no output, not counted."""
        B, k, V = draft_probs.shape
        dev = draft_probs.device
        if seeded_seqs:
            if uniform is not None or exponential is not None:
                raise ValueError("seeded_seqs and injected draws are mutually exclusive")
            uniform = torch.rand(B, k, device=dev, dtype=torch.float32)
            exponential = torch.empty(B, k, V, device=dev, dtype=torch.float32).exponential_(1.0)
            for row, gen in seeded_seqs.items():
                uniform[row] = torch.rand(1, k, device=dev, dtype=torch.float32, generator=gen)
                exponential[row] = torch.empty(k, V, device=dev, dtype=torch.float32).exponential_(1.0, generator=gen)
        if self._strict_mode:
            self._raise_if_incorrect_input(target_with_bonus_probs, draft_token_ids, bonus_token_ids, draft_probs)
        if B == 0:
            return torch.empty(0, k + 1, device=dev, dtype=torch.int64)
        out = out if out is not None else torch.empty(B, k + 1, dtype=torch.int64, device=dev)
        accepted = accepted if accepted is not None else torch.empty(B, k, dtype=torch.uint8, device=dev)
        recovered = recovered if recovered is not None else torch.empty(B, k, dtype=torch.int64, device=dev)
        bonus = bonus_token_ids.squeeze(-1) if bonus_token_ids.dim() == 2 else bonus_token_ids  # a view, never a copy
        ops.rejection_sample(target_with_bonus_probs, bonus, draft_probs, draft_token_ids, out,
                             accepted, recovered, self.counters, uniform=uniform, exponential=exponential,
                             rng_state=self.rng_state, active_lens=active_lens)
        return out

    __call__ = forward

    # spec_decode_base_sampler.py:133-254
    def _raise_if_incorrect_input(self, target_with_bonus_probs, draft_token_ids, bonus_token_ids, draft_probs):
        B, k1, V = target_with_bonus_probs.shape
        assert draft_probs.shape == (B, k1 - 1, V), "draft/target shape mismatch"
        assert draft_token_ids.shape == (B, k1 - 1)
        assert bonus_token_ids.numel() == B
        assert target_with_bonus_probs.dtype == draft_probs.dtype == torch.float32
        assert draft_token_ids.dtype == bonus_token_ids.dtype == torch.int64
        devs = {t.device for t in (target_with_bonus_probs, draft_probs, draft_token_ids, bonus_token_ids)}
        assert len(devs) == 1
        assert int(draft_token_ids.max()) < V and int(draft_token_ids.min()) >= 0
        assert int(bonus_token_ids.max()) < V and int(bonus_token_ids.min()) >= 0
