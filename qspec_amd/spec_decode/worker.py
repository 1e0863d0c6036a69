"""Worker boundary: what vLLM's executor calls.

`create_spec_worker(**kwargs) -> SpecDecodeWorker` and the methods vLLM's WorkerBase contract needs
(`init_device`, `load_model`, `determine_num_available_blocks`, `initialize_cache`, `execute_model`,
`get_cache_block_size_bytes`, `start_worker_execution_loop`, `rank`, `device`), mirroring
vllm/spec_decode/spec_decode_worker.py:53-113,118-470,461-560.  vLLM itself is not importable here, so the
request / output records are small dataclasses with the reference's field names (vllm/sequence.py).

QSpec specifics preserved:
  * proposer and scorer are the SAME model object and the SAME KV cache (:339-345, :421-444);
  * `execute_model_req.w4a4 = True` only around the proposer (:797-812); `ExecuteModelRequest.clone()` does not
    carry `w4a4`, so the scorer always runs W4A16 (vllm/sequence.py:1301,1331-1348);
  * prefill never runs the proposer (:699);
  * one `SamplerOutput` per emitted position, `-1` = no token for that sequence (:972-1063).
The decisions the reference inherits from `attn_backend.get_name() == "FLASH_ATTN"` (MQA scorer, on-GPU draft
loop; :214-235, draft_model_runner.py:154) are owned here: both are always on.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from ..model import CONFIGS, QuarotLlamaConfig, QuarotLlamaForCausalLM
from .engine import QSpecEngine
from .metrics import AsyncMetricsCollector, SpecDecodeWorkerMetrics


@dataclass
class SequenceData:
    prompt_token_ids: List[int]
    output_token_ids: List[int] = field(default_factory=list)

    def get_len(self):
        return len(self.prompt_token_ids) + len(self.output_token_ids)


@dataclass
class SequenceGroupMetadata:
    request_id: str
    is_prompt: bool
    seq_data: Dict[int, SequenceData]
    num_speculative_tokens: Optional[int] = None


@dataclass
class ExecuteModelRequest:
    seq_group_metadata_list: List[SequenceGroupMetadata]
    num_lookahead_slots: int = 0
    running_queue_size: int = 0
    w4a4: bool = False                                   # vllm/sequence.py:1301

    def clone(self, seq_group_metadata_list):
        """vllm/sequence.py:1331-1348: every field except `w4a4` is carried over."""
        return ExecuteModelRequest(seq_group_metadata_list=seq_group_metadata_list,
                                   num_lookahead_slots=self.num_lookahead_slots,
                                   running_queue_size=self.running_queue_size)


@dataclass
class SamplerOutput:
    """One decode position for the whole batch (vllm/model_executor/layers/sampler.py SamplerOutput)."""
    sampled_token_ids: torch.Tensor                      # [B] int64 on the host, -1 = nothing emitted
    request_ids: List[str]
    spec_decode_worker_metrics: Optional[SpecDecodeWorkerMetrics] = None


@dataclass
class SpeculativeConfig:
    num_speculative_tokens: int = 3
    speculative_disable_mqa_scorer: bool = False
    speculative_disable_by_batch_size: Optional[int] = None
    draft_token_acceptance_method: str = "rejection_sampler"


def create_spec_worker(*args, **kwargs) -> "SpecDecodeWorker":
    """Entry point resolved from `parallel_config.worker_cls` (vllm/platforms/rocm.py:130-131).

    kwargs: model_config (QuarotLlamaConfig or a CONFIGS name), speculative_config, max_num_seqs,
    max_model_len, block_size, device, model (optional pre-built QuarotLlamaForCausalLM), seed,
    pipeline_parallel_size."""
    if kwargs.get("pipeline_parallel_size", 1) > 1:
        raise NotImplementedError("Speculative decoding is currently incompatible with pipeline parallelism")
    spec = kwargs.get("speculative_config") or SpeculativeConfig()
    if spec.draft_token_acceptance_method != "rejection_sampler":
        raise NotImplementedError("only the rejection sampler is on the QSpec path")
    cfg = kwargs.get("model_config", "llama-3-8b")
    if isinstance(cfg, str):
        cfg = CONFIGS[cfg]
    return SpecDecodeWorker(cfg, spec, max_num_seqs=kwargs.get("max_num_seqs", 4),
                            max_model_len=kwargs.get("max_model_len", 1024), block_size=kwargs.get("block_size", 16),
                            device=kwargs.get("device", "cuda:0"), model=kwargs.get("model"),
                            seed=kwargs.get("seed", 0), rank=kwargs.get("rank", 0),
                            disable_log_stats=kwargs.get("disable_log_stats", False))


class SpecDecodeWorker:
    def __init__(self, model_config: QuarotLlamaConfig, speculative_config: SpeculativeConfig, max_num_seqs: int,
                 max_model_len: int, block_size: int, device: str, model: Optional[QuarotLlamaForCausalLM] = None,
                 seed: int = 0, rank: int = 0, disable_log_stats: bool = False):
        self.model_config = model_config
        self.speculative_config = speculative_config
        self.max_num_seqs = max_num_seqs
        self.max_model_len = max_model_len
        self.block_size = block_size
        self._device = torch.device(device)
        self._model = model
        self._seed = seed
        self._rank = rank
        self._driver_rank = 0
        self.disable_by_batch_size = speculative_config.speculative_disable_by_batch_size
        self._disable_log_stats = disable_log_stats
        self.engine: Optional[QSpecEngine] = None
        self._metrics: Optional[AsyncMetricsCollector] = None
        self._slots: Dict[str, int] = {}
        self.proposer_calls = 0     # forwards run with w4a4=True (for tests of the toggle)
        self.scorer_calls = 0

    # ------------------------------------------------------------------ WorkerBase contract
    @property
    def rank(self):
        return self._rank

    @property
    def device(self):
        return self._device

    def init_device(self) -> None:
        """:326-369: the scorer loads the model, the proposer receives the very same object."""
        torch.cuda.set_device(self._device)
        if self._model is None:
            self._model = QuarotLlamaForCausalLM(self.model_config, self._device).init_synthetic(self._seed)
        self.scorer_model = self._model
        self.proposer_model = self._model      # load_model(self.scorer_worker.model_runner.model), :342

    def load_model(self, *args, **kwargs):
        pass                                   # :371

    def get_model(self):
        return self._model

    def determine_num_available_blocks(self) -> Tuple[int, int]:
        """:400-426: the scorer's block count, NOT split between proposer and scorer (shared cache)."""
        blocks_per_seq = (self.max_model_len + self.block_size - 1) // self.block_size
        return self.max_num_seqs * blocks_per_seq, 0

    def initialize_cache(self, num_gpu_blocks: int, num_cpu_blocks: int) -> None:
        """:428-444 + vllm/worker/worker.py:309-327 (ref_initilize_cache): one cache engine for both workers."""
        blocks_per_seq = (self.max_model_len + self.block_size - 1) // self.block_size
        assert num_gpu_blocks >= self.max_num_seqs * blocks_per_seq
        self.engine = QSpecEngine(self._model, self.speculative_config.num_speculative_tokens, self.max_num_seqs,
                                  self.max_model_len, self.block_size, seed=self._seed)
        self._metrics = AsyncMetricsCollector(self.engine.sampler)
        self._metrics.init_gpu_tensors(self._rank)

    def get_cache_block_size_bytes(self):
        raise NotImplementedError  # as the reference (:1259-1268)

    def start_worker_execution_loop(self) -> None:
        raise NotImplementedError("non-driver ranks are driven by qspec_amd.parallel (TP), not by a broadcast loop")

    def start_profile(self):
        torch.cuda.profiler.start()

    def stop_profile(self):
        torch.cuda.profiler.stop()

    # ------------------------------------------------------------------ execute_model (:461-538)
    @torch.inference_mode()
    def execute_model(self, execute_model_req: Optional[ExecuteModelRequest] = None) -> List[SamplerOutput]:
        if execute_model_req is None:
            return []
        sgml = execute_model_req.seq_group_metadata_list
        assert sgml is not None, "speculative decoding requires non-None seq_group_metadata_list"
        num_lookahead_slots = execute_model_req.num_lookahead_slots
        all_prompt = all(s.is_prompt for s in sgml)
        all_zero_spec = all(s.num_speculative_tokens == 0 for s in sgml)
        if all_prompt and sgml:
            assert num_lookahead_slots == 0, "Prompt only runs should have num_lookahead_slots equal to 0."
        disable_all_speculation = self._should_disable_all_speculation(execute_model_req)
        no_spec = num_lookahead_slots == 0 or disable_all_speculation or all_zero_spec
        if no_spec:
            return self._run_no_spec(execute_model_req, skip_proposer=True)      # skip_proposer forced (:699)
        return self._run_speculative_decoding_step(execute_model_req, num_lookahead_slots)

    def _should_disable_all_speculation(self, req: ExecuteModelRequest) -> bool:
        return self.disable_by_batch_size is not None and req.running_queue_size >= self.disable_by_batch_size

    def _request_ids(self, sgml):
        return [s.request_id for s in sgml]

    def _run_no_spec(self, req: ExecuteModelRequest, skip_proposer: bool) -> List[SamplerOutput]:
        """:666-720.  Prompts: W4A16 prefill, first token sampled by the target."""
        sgml = req.seq_group_metadata_list
        assert req.w4a4 is False
        if not any(s.is_prompt for s in sgml):
            # decode batch with speculation off for this step: the scorer alone emits one token per sequence
            assert [self._slots[s.request_id] for s in sgml] == list(range(len(sgml))), "the batch is fixed after prefill"
            self.engine.step_no_spec()
            self.scorer_calls += 1
            return [SamplerOutput(self.engine.out_tokens[:, 0].cpu(), self._request_ids(sgml))]
        if not all(s.is_prompt for s in sgml):
            raise NotImplementedError("mixed prompt / decode batches are not scheduled onto this worker")
        prompts = [next(iter(s.seq_data.values())).prompt_token_ids for s in sgml]
        self.engine.add_sequences(prompts)
        self.scorer_calls += 1
        self._slots = {s.request_id: i for i, s in enumerate(sgml)}
        first = self.engine.gen_tokens[:, 0].cpu()
        return [SamplerOutput(first, self._request_ids(sgml))]

    def _run_speculative_decoding_step(self, req: ExecuteModelRequest, num_lookahead_slots: int) -> List[SamplerOutput]:
        """:758-858 as one graph replay: proposals (w4a4=True) -> scoring (w4a4=False) -> verification."""
        k = self.engine.k
        assert num_lookahead_slots == k, "the cycle graph is captured for a fixed k"
        sgml = req.seq_group_metadata_list
        req.w4a4 = True                                   # :799
        self.proposer_calls += k
        scorer_req = req.clone(sgml)                      # mqa_scorer.py:65: clone drops w4a4 -> W4A16
        assert scorer_req.w4a4 is False
        req.w4a4 = False                                  # :812
        self.scorer_calls += 1
        self.engine.step()
        return self._create_output_sampler_list(sgml, k)

    def _create_output_sampler_list(self, sgml, k: int) -> List[SamplerOutput]:
        """:972-1063: transpose [B, k+1] -> k+1 per-step outputs; metrics ride on the first one."""
        out = self.engine.out_tokens.cpu()                # the one host sync of the cycle (reference: three)
        ids = self._request_ids(sgml)
        order = [self._slots[r] for r in ids]
        outs = [SamplerOutput(out[order, j].clone(), ids) for j in range(k + 1)]
        # drop trailing steps in which no sequence emitted anything (:1038-1046)
        while len(outs) > 1 and bool((outs[-1].sampled_token_ids == -1).all()):
            outs.pop()
        if not self._disable_log_stats:
            outs[0].spec_decode_worker_metrics = self._metrics.maybe_collect_rejsample_metrics(k)
        return outs
