"""Worker boundary: what vLLM's executor calls.

`create_spec_worker(*args, **kwargs) -> SpecDecodeWorker` and the methods vLLM's WorkerBase contract needs
(`init_device`, `load_model`, `determine_num_available_blocks`, `initialize_cache`, `execute_model`,
`get_cache_block_size_bytes`, `start_worker_execution_loop`, `rank`, `device`), mirroring
vllm/spec_decode/spec_decode_worker.py:53-113,118-470,461-560,582-664,722-755,972-1110,1178-1210.

The factory binds the way the reference's does: `kwargs["vllm_config"]` (duck-typed: `model_config.{hf_config,
max_model_len, model, seed, max_logprobs}`, `cache_config.{block_size, gpu_memory_utilization, swap_space_bytes}`,
`scheduler_config.{max_num_seqs, max_num_batched_tokens}`, `speculative_config.{num_speculative_tokens,
speculative_disable_by_batch_size, disable_log_stats, disable_logprobs, draft_token_acceptance_method}`,
`parallel_config.{tensor_parallel_size, pipeline_parallel_size}`, `load_config.load_format`) plus `local_rank`, `rank`,
`distributed_init_method`, `is_driver_worker` (:53-113; vllm/worker/worker.py:46-60).  A second, explicit-keyword path
(`model_config=`, `max_num_seqs=`, ...) is kept for tests and benchmarks.

vLLM itself is not importable here, so the request / output records are small dataclasses with the reference's
field names (vllm/sequence.py:37-48,1017-1078; vllm/model_executor/layers/sampler.py SamplerOutput): what
`MultiStepOutputProcessor` reads -- `output.samples[0].output_token`, `.parent_seq_id`, `.logprobs`
(vllm/engine/output_processor/multi_step.py:100-176) -- is there under those names.

QSpec specifics preserved:
  * proposer and scorer are the SAME model object and the SAME KV cache (:339-345, :421-444); the block count is the
    scorer's, not split between the two (:421-423);
  * `execute_model_req.w4a4 = True` only around the proposer (:797-812); `ExecuteModelRequest.clone()` does not
    carry `w4a4`, so the scorer always runs W4A16 (vllm/sequence.py:1301,1331-1348);
  * prefill never runs the proposer (:699);
  * one `SamplerOutput` per emitted position, `-1` = no token for that sequence, the list stops at the first position
    where no sequence emitted anything (:1023-1026);
  * bonus-token bookkeeping is kept (:1178-1210) although `llama_quarot` skips the batch expansion it feeds
    (multi_step_worker.py:74-80): the verify pass has already written the KV of every accepted position.
The decisions the reference inherits from `attn_backend.get_name() == "FLASH_ATTN"` (MQA scorer, on-GPU draft
loop; :214-235, draft_model_runner.py:154) are owned here: both are always on.

Batch membership follows `seq_group_metadata_list` on EVERY call: requests are identified by `request_id`, keep their
engine slot until `finished_requests_ids` names them, may be scheduled in any order and in any subset (sequences that
sit out a step do not advance), and new prompts are admitted whenever a slot is free.  The scheduler's block tables
(`SequenceGroupMetadata.block_tables`) are honoured when given.

Tensor parallelism (the verify pass shards, qspec_amd/parallel.py): the driver rank broadcasts the step's control
record and the request, non-driver ranks sit in `start_worker_execution_loop()` and mirror the call
(:524-538, :722-755; vllm/worker/worker_base.py:340-344).
"""
from __future__ import annotations

import os
from collections import defaultdict
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional, Set, Tuple

import torch

from ..model import CONFIGS, QuarotLlamaConfig, QuarotLlamaForCausalLM
from .engine import QSpecEngine
from .metrics import AsyncMetricsCollector, SpecDecodeWorkerMetrics

VLLM_INVALID_TOKEN_ID = -1       # vllm/sequence.py:29


# ---------------------------------------------------------------------- records (vllm/sequence.py field names)

@dataclass
class SamplingParams:
    """The fields of vllm/sampling_params.py the worker and the output processor read."""
    logprobs: Optional[int] = None
    prompt_logprobs: Optional[int] = None
    max_tokens: int = 16
    ignore_eos: bool = False
    seed: Optional[int] = None
    # Sampler.forward's inputs (sampler.py:216-316).  temperature 0 = greedy -- the reference's demo and BASELINE.json's configs;
    # vLLM's own default is 1.0, a caller coming from vLLM passes its SamplingParams object and with it the real value
    temperature: float = 0.0
    top_k: int = -1
    top_p: float = 1.0


@dataclass
class SequenceData:
    prompt_token_ids: List[int]
    output_token_ids: List[int] = field(default_factory=list)

    def get_len(self):
        return len(self.prompt_token_ids) + len(self.output_token_ids)

    def get_output_len(self):
        return len(self.output_token_ids)

    def get_prompt_token_ids(self):
        return list(self.prompt_token_ids)

    def get_token_ids(self):
        return list(self.prompt_token_ids) + list(self.output_token_ids)


@dataclass
class SequenceGroupMetadata:
    request_id: str
    is_prompt: bool
    seq_data: Dict[int, SequenceData]
    num_speculative_tokens: Optional[int] = None
    block_tables: Optional[Dict[int, List[int]]] = None   # seq_id -> physical block ids (vllm/sequence.py)
    sampling_params: Optional[SamplingParams] = None
    do_sample: bool = True
    token_chunk_size: Optional[int] = None


@dataclass
class ExecuteModelRequest:
    seq_group_metadata_list: List[SequenceGroupMetadata]
    num_lookahead_slots: int = 0
    running_queue_size: int = 0
    finished_requests_ids: List[str] = field(default_factory=list)   # vllm/sequence.py:1295
    w4a4: bool = False                                   # vllm/sequence.py:1301

    def clone(self, seq_group_metadata_list):
        """vllm/sequence.py:1331-1348: every field except `w4a4` is carried over."""
        return ExecuteModelRequest(seq_group_metadata_list=seq_group_metadata_list,
                                   num_lookahead_slots=self.num_lookahead_slots,
                                   running_queue_size=self.running_queue_size,
                                   finished_requests_ids=self.finished_requests_ids)


@dataclass
class Logprob:
    """vllm/sequence.py:37-48."""
    logprob: float
    rank: Optional[int] = None
    decoded_token: Optional[str] = None


@dataclass
class SequenceOutput:
    """vllm/sequence.py:1017-1045."""
    parent_seq_id: int
    output_token: int
    logprobs: Dict[int, Logprob]


@dataclass
class CompletionSequenceGroupOutput:
    """vllm/sequence.py:1060-1078."""
    samples: List[SequenceOutput]
    prompt_logprobs: Optional[List[Optional[Dict[int, Logprob]]]] = None


@dataclass
class SamplerOutput:
    """One decode position for the whole batch (vllm/model_executor/layers/sampler.py SamplerOutput): `outputs[i]` is
    the i-th sequence group of the request, in request order.  The device-side fields are cleared before the list
    leaves the worker, as the reference does (:715-719)."""
    outputs: List[CompletionSequenceGroupOutput]
    sampled_token_probs: Optional[torch.Tensor] = None
    sampled_token_ids: Optional[torch.Tensor] = None
    logprobs: Optional[torch.Tensor] = None
    spec_decode_worker_metrics: Optional[SpecDecodeWorkerMetrics] = None
    hidden_states: Optional[torch.Tensor] = None

    def __getitem__(self, idx: int) -> CompletionSequenceGroupOutput:
        return self.outputs[idx]

    def __len__(self):
        return len(self.outputs)

    def token_ids(self) -> List[int]:
        """Convenience (tests, benchmarks): the output token per sequence group, -1 where a group has none."""
        return [o.samples[0].output_token if o.samples else VLLM_INVALID_TOKEN_ID for o in self.outputs]


@dataclass
class SpeculativeConfig:
    num_speculative_tokens: int = 3
    speculative_disable_mqa_scorer: bool = False
    speculative_disable_by_batch_size: Optional[int] = None
    draft_token_acceptance_method: str = "rejection_sampler"       # or "typical_acceptance_sampler"
    typical_acceptance_sampler_posterior_threshold: float = 0.09    # vllm/config.py defaults
    typical_acceptance_sampler_posterior_alpha: float = 0.3
    disable_logprobs: bool = True        # vllm/config.py:1788-1789: logprobs are off during speculation by default
    disable_log_stats: bool = False


class DeviceHandoffTimeout(RuntimeError):
    """A device-side hand-off (spread Hadamard exchange, norm hand-off, one-shot all-reduce) timed out and the recovery
    replay without hand-offs failed as well: the cycle's results are invalid on every rank."""


def create_logprobs_output(token_id: int, token_id_logprob_rank: int, token_id_logprob: float,
                           topk_token_ids: List[Optional[int]], topk_logprobs: List[Optional[float]]) -> Dict[int, Logprob]:
    """vllm/spec_decode/util.py:54-88."""
    logprobs = {token_id: Logprob(logprob=token_id_logprob, rank=token_id_logprob_rank)}
    logprobs.update({tid: Logprob(logprob=lp if lp is not None else 0.0, rank=i + 1)
                     for i, (tid, lp) in enumerate(zip(topk_token_ids, topk_logprobs)) if tid is not None})
    return logprobs


def create_sequence_group_output(token_id: int, token_id_logprob_rank: int, token_id_logprob: float, seq_id: int,
                                 topk_token_ids: List[Optional[int]], topk_logprobs: List[Optional[float]],
                                 prompt_logprobs=None) -> CompletionSequenceGroupOutput:
    """vllm/spec_decode/util.py:91-126."""
    logprobs = create_logprobs_output(token_id, token_id_logprob_rank, token_id_logprob, topk_token_ids, topk_logprobs)
    return CompletionSequenceGroupOutput(
        samples=[SequenceOutput(parent_seq_id=seq_id, output_token=token_id, logprobs=logprobs)],
        prompt_logprobs=prompt_logprobs)


# ---------------------------------------------------------------------- factory (:53-113)

def model_config_from_hf(hf_config, name: str = "hf") -> QuarotLlamaConfig:
    """HF LlamaConfig-shaped object -> the fields this path needs (quarot_llama.py:319-360,436-470)."""
    g = lambda k, d=None: getattr(hf_config, k, d)  # noqa: E731
    return QuarotLlamaConfig(g("hidden_size"), g("intermediate_size"), g("num_attention_heads"),
                             g("num_key_value_heads", g("num_attention_heads")), g("num_hidden_layers"),
                             g("vocab_size"), g("rms_norm_eps", 1e-5), float(g("rope_theta", 10000.0)),
                             g("max_position_embeddings", 8192), g("_name_or_path", name) or name)


def create_spec_worker(*args, **kwargs) -> "SpecDecodeWorker":
    """Entry point resolved from `parallel_config.worker_cls` (vllm/platforms/rocm.py:130-131); reference :53-113.

    Path 1 (what vLLM's WorkerWrapper passes): kwargs = {vllm_config, local_rank, rank, distributed_init_method,
    is_driver_worker}.  Path 2 (tests, benchmarks): model_config (QuarotLlamaConfig or a CONFIGS name),
    speculative_config, max_num_seqs, max_model_len, block_size, device, model (a pre-built
    QuarotLlamaForCausalLM), seed, pipeline_parallel_size, rank, engine_factory."""
    vllm_config = kwargs.get("vllm_config")
    if vllm_config is not None:
        spec = vllm_config.speculative_config
        assert spec is not None                                                     # :59
        par = vllm_config.parallel_config
        if getattr(par, "pipeline_parallel_size", 1) > 1:                           # :61-63
            raise NotImplementedError("Speculative decoding is currently incompatible with pipeline parallelism")
        method = getattr(spec, "draft_token_acceptance_method", "rejection_sampler")
        if method not in ("rejection_sampler", "typical_acceptance_sampler"):       # :95-110
            raise ValueError(f"draft_token_acceptance_method {method!r}: expected rejection_sampler or typical_acceptance_sampler")
        mc, cc, sc = vllm_config.model_config, vllm_config.cache_config, vllm_config.scheduler_config
        model_type = getattr(mc.hf_config, "model_type", "llama_quarot")
        if model_type not in ("llama_quarot", "llama"):
            raise NotImplementedError(f"model_type {model_type!r}: this worker runs the llama_quarot QSpec model only "
                                      "(vllm/worker/model_runner.py:1104-1148)")
        local_rank = kwargs.get("local_rank", 0)
        load_format = getattr(getattr(vllm_config, "load_config", None), "load_format", "auto")
        return SpecDecodeWorker(
            model_config_from_hf(mc.hf_config, str(getattr(mc, "model", "hf"))), spec,
            max_num_seqs=sc.max_num_seqs, max_model_len=mc.max_model_len, block_size=cc.block_size,
            device=kwargs.get("device", f"cuda:{local_rank}"), model=kwargs.get("model"),
            seed=getattr(mc, "seed", 0) or 0, rank=kwargs.get("rank", 0),
            disable_log_stats=getattr(spec, "disable_log_stats", False), engine_factory=kwargs.get("engine_factory"),
            model_path=getattr(mc, "model", None), load_format=str(load_format),
            tensor_parallel_size=getattr(par, "tensor_parallel_size", 1),
            distributed_init_method=kwargs.get("distributed_init_method"),
            is_driver_worker=kwargs.get("is_driver_worker", kwargs.get("rank", 0) == 0),
            gpu_memory_utilization=getattr(cc, "gpu_memory_utilization", 0.9),
            swap_space_bytes=getattr(cc, "swap_space_bytes", 0),
            max_num_batched_tokens=getattr(sc, "max_num_batched_tokens", None),
            max_logprobs=getattr(mc, "max_logprobs", 20),
            disable_logprobs=getattr(spec, "disable_logprobs", True))
    if kwargs.get("pipeline_parallel_size", 1) > 1:
        raise NotImplementedError("Speculative decoding is currently incompatible with pipeline parallelism")
    spec = kwargs.get("speculative_config") or SpeculativeConfig()
    if spec.draft_token_acceptance_method not in ("rejection_sampler", "typical_acceptance_sampler"):
        raise ValueError(f"draft_token_acceptance_method {spec.draft_token_acceptance_method!r}: expected rejection_sampler or "
                         "typical_acceptance_sampler")
    cfg = kwargs.get("model_config", "llama-3-8b")
    if isinstance(cfg, str):
        cfg = CONFIGS[cfg]
    return SpecDecodeWorker(cfg, spec, max_num_seqs=kwargs.get("max_num_seqs", 4),
                            max_model_len=kwargs.get("max_model_len", 1024), block_size=kwargs.get("block_size", 16),
                            device=kwargs.get("device", "cuda:0"), model=kwargs.get("model"),
                            seed=kwargs.get("seed", 0), rank=kwargs.get("rank", 0),
                            disable_log_stats=kwargs.get("disable_log_stats", getattr(spec, "disable_log_stats", False)),
                            engine_factory=kwargs.get("engine_factory"), model_path=kwargs.get("model_path"),
                            load_format=kwargs.get("load_format", "dummy" if kwargs.get("model_path") is None else "auto"),
                            gpu_memory_utilization=kwargs.get("gpu_memory_utilization", 0.9),
                            max_num_batched_tokens=kwargs.get("max_num_batched_tokens"),
                            max_logprobs=kwargs.get("max_logprobs", 20),
                            disable_logprobs=getattr(spec, "disable_logprobs", True),
                            memory_probe=kwargs.get("memory_probe"))


class SpecDecodeWorker:
    def __init__(self, model_config: QuarotLlamaConfig, speculative_config, max_num_seqs: int,
                 max_model_len: int, block_size: int, device: str, model: Optional[QuarotLlamaForCausalLM] = None,
                 seed: int = 0, rank: int = 0, disable_log_stats: bool = False, engine_factory=None,
                 model_path: Optional[str] = None, load_format: str = "auto", tensor_parallel_size: int = 1,
                 distributed_init_method: Optional[str] = None, is_driver_worker: bool = True,
                 gpu_memory_utilization: float = 0.9, swap_space_bytes: int = 0,
                 max_num_batched_tokens: Optional[int] = None, max_logprobs: int = 20, disable_logprobs: bool = True,
                 memory_probe: Optional[Callable[[], Tuple[int, int, int]]] = None):
        self.model_config = model_config
        self.speculative_config = speculative_config
        self.max_num_seqs = max_num_seqs
        self.max_model_len = max_model_len
        self.block_size = block_size
        self._device = torch.device(device)
        self._model = model
        self._model_path, self._load_format = model_path, load_format
        self._seed = seed
        self._rank = rank
        self._driver_rank = 0
        self._is_driver_worker = is_driver_worker
        self._tp_size, self._dist_init = tensor_parallel_size, distributed_init_method
        self.gpu_memory_utilization, self.swap_space_bytes = gpu_memory_utilization, swap_space_bytes
        self.max_num_batched_tokens = max_num_batched_tokens
        self.max_logprobs = max_logprobs
        self._disable_logprobs = disable_logprobs
        self._memory_probe = memory_probe
        self.disable_by_batch_size = getattr(speculative_config, "speculative_disable_by_batch_size", None)
        self._disable_log_stats = disable_log_stats
        self._engine_factory = engine_factory or QSpecEngine
        self.engine: Optional[QSpecEngine] = None
        self._metrics: Optional[AsyncMetricsCollector] = None
        self._slots: Dict[str, int] = {}                 # request_id -> engine slot
        # :1178-1210 -- sequences that received a bonus token in their last step / request -> its sequence ids
        self._seq_with_bonus_token_in_last_step: Set[int] = set()
        self._request_id_seq_id_mapping: Dict[str, Set[int]] = defaultdict(set)
        self.proposer_calls = 0     # forwards run with w4a4=True (for tests of the toggle)
        self.scorer_calls = 0
        self.memory_profile: Dict[str, Any] = {}

    # ------------------------------------------------------------------ WorkerBase contract
    @property
    def rank(self):
        return self._rank

    @property
    def device(self):
        return self._device

    @property
    def _tp(self):
        tp = getattr(self._model, "tp", None)
        return tp if tp is not None and tp.world > 1 else None

    def init_device(self) -> None:
        """:326-369: the scorer initialises the device and loads the model, the proposer receives the very same
        object (`proposer_worker.load_model(self.scorer_worker.model_runner.model)`, :342)."""
        if self._device.type == "cuda":
            torch.cuda.set_device(self._device)
        if self._tp_size > 1 and self._model is None:
            import torch.distributed as dist
            if not dist.is_initialized():     # vllm/worker/worker.py:init_worker_distributed_environment
                dist.init_process_group("nccl", init_method=self._dist_init, world_size=self._tp_size, rank=self._rank,
                                        device_id=self._device)
        if self._model is None:
            self._model = self._load_model()
        self.scorer_model = self._model
        self.proposer_model = self._model

    def _load_model(self) -> QuarotLlamaForCausalLM:
        """vllm/worker/model_runner.py:1096-1148: `model_config.model` is a directory holding the two safetensors
        shards of a QSpec checkpoint (loaded through qspec_amd/checkpoint.py: key renames, fuse_qkv / fuse_gate_up);
        `load_format == "dummy"` (vLLM's random-weights format) gives the SURVEY 8d synthetic weights.  Anything else
        raises: a worker must never run on silently invented weights."""
        model = QuarotLlamaForCausalLM(self.model_config, self._device)
        if self._load_format == "dummy":
            model.init_synthetic(self._seed)
        elif self._model_path is not None and os.path.isdir(str(self._model_path)):
            from ..checkpoint import load_qspec_checkpoint
            load_qspec_checkpoint(model, str(self._model_path))
        else:
            raise FileNotFoundError(f"model_config.model = {self._model_path!r} is not a local QSpec checkpoint directory "
                                    "(no hub access here); pass load_format='dummy' for synthetic weights")
        if self._tp_size > 1:
            from ..parallel import attach_tp
            k = self.speculative_config.num_speculative_tokens
            attach_tp(model, self._rank, self._tp_size, tokens=self.max_num_seqs * (k + 1))
        return model

    def load_model(self, *args, **kwargs):
        pass                                   # :371 (the model is loaded in init_device, as in the reference)

    def get_model(self):
        return self._model

    def cache_block_size_bytes(self) -> int:
        """vllm/worker/cache_engine.py:101-119: K and V of one block over all layers, fp16.  The KV cache is NOT
        sharded under this worker's tensor parallelism (replicated draft pass), so kv heads are not divided."""
        c = self.model_config
        return 2 * c.num_hidden_layers * self.block_size * c.num_key_value_heads * c.head_dim * 2

    def _profile_memory(self) -> Tuple[int, int, int]:
        """(free bytes now, total bytes, peak increase of one profiled prompt pass + cycle).  vllm/worker/worker.py:176-205:
        `profile_run()` under `memory_profiling`: here a throw-away engine with a KV cache just large enough for the
        profiled batch runs the worst-case prompt pass (max_num_batched_tokens over max_num_seqs prompts) and one
        captured cycle; torch's peak allocation above the weights is what the cycle needs besides the KV cache."""
        if self._memory_probe is not None:
            return self._memory_probe()
        if self._device.type != "cuda":
            raise RuntimeError("memory profiling needs the GPU (pass memory_probe= for host-side tests)")
        import gc
        dev = self._device
        gc.collect()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(dev)
        base = torch.cuda.memory_allocated(dev)
        k = self.speculative_config.num_speculative_tokens
        budget = self.max_num_batched_tokens or self.max_model_len
        per_seq = max(1, min(self.max_model_len - (k + 2), budget // self.max_num_seqs))
        blocks_per_seq = (per_seq + k + 2 + self.block_size - 1) // self.block_size
        n_blocks = self.max_num_seqs * blocks_per_seq
        eng = self._engine_factory(self._model, k, self.max_num_seqs, self.max_model_len, self.block_size,
                                   seed=self._seed, num_blocks=n_blocks, **self._sampler_kwargs())
        kv_bytes = n_blocks * self.cache_block_size_bytes()
        tables = [list(range(b * blocks_per_seq, (b + 1) * blocks_per_seq)) for b in range(self.max_num_seqs)]
        eng.add_sequences_to(list(range(self.max_num_seqs)), [[0] * per_seq] * self.max_num_seqs, tables)
        eng.step()
        torch.cuda.synchronize(dev)
        peak = torch.cuda.max_memory_allocated(dev) - base - kv_bytes
        del eng
        gc.collect()
        torch.cuda.empty_cache()
        free, total = torch.cuda.mem_get_info(dev)
        return int(free), int(total), int(max(peak, 0))

    def determine_num_available_blocks(self) -> Tuple[int, int]:
        """:400-426 over vllm/worker/worker.py:176-265: profile the scorer, then
            available = total x gpu_memory_utilization - non_kv_cache_memory - avoid_oom_memory,
        with non_kv_cache_memory = weights + non-torch + torch peak increase (vllm/utils.py:2053-2056) and the fork's
        head-room for speculative decoding avoid_oom_memory = 2 x torch peak increase (worker.py:227-231); the count is
        NOT split between proposer and scorer (shared cache, :421-423)."""
        free, total, peak = self._profile_memory()
        used_without_kv = total - free                    # weights + non-torch + whatever else lives on the card
        avoid_oom = 2 * peak                              # worker.py:227-231 (speculative_config is not None here)
        available = total * self.gpu_memory_utilization - (used_without_kv + peak) - avoid_oom
        block = self.cache_block_size_bytes()
        num_gpu_blocks = max(int(available // block), 0)
        num_cpu_blocks = max(int(self.swap_space_bytes // block), 0)
        self.memory_profile = dict(free=free, total=total, torch_peak_increase=peak, avoid_oom_memory=avoid_oom,
                                   available_kv_cache_memory=int(available), cache_block_size=block)
        return num_gpu_blocks, num_cpu_blocks

    def initialize_cache(self, num_gpu_blocks: int, num_cpu_blocks: int) -> None:
        """:428-444 + vllm/worker/worker.py:290-327 (raise_if_cache_size_invalid, ref_initilize_cache): one cache
        engine of `num_gpu_blocks` blocks for both passes."""
        if num_gpu_blocks <= 0:
            raise ValueError("No available memory for the cache blocks. Try increasing `gpu_memory_utilization` when "
                             "initializing the engine.")
        if self.max_model_len > self.block_size * num_gpu_blocks:
            raise ValueError(f"The model's max seq len ({self.max_model_len}) is larger than the maximum number of "
                             f"tokens that can be stored in KV cache ({self.block_size * num_gpu_blocks}). Try "
                             "increasing `gpu_memory_utilization` or decreasing `max_model_len` when initializing the engine.")
        self.engine = self._engine_factory(self._model, self.speculative_config.num_speculative_tokens,
                                           self.max_num_seqs, self.max_model_len, self.block_size, seed=self._seed,
                                           num_blocks=num_gpu_blocks, **self._sampler_kwargs())
        self._metrics = AsyncMetricsCollector(self.engine.sampler)
        self._metrics.init_gpu_tensors(self._rank)

    def _sampler_kwargs(self) -> dict:
        """spec_decode_worker.py:95-110: the acceptance sampler named by `draft_token_acceptance_method`."""
        spec = self.speculative_config
        if getattr(spec, "draft_token_acceptance_method", "rejection_sampler") != "typical_acceptance_sampler":
            return {}
        from .typical_acceptance_sampler import TypicalAcceptanceSampler
        return {"acceptance_sampler": TypicalAcceptanceSampler(
            posterior_threshold=getattr(spec, "typical_acceptance_sampler_posterior_threshold", 0.09),
            posterior_alpha=getattr(spec, "typical_acceptance_sampler_posterior_alpha", 0.3), seed=self._seed)}

    def get_cache_block_size_bytes(self):
        raise NotImplementedError  # as the reference (:1259-1268)

    def start_profile(self):
        torch.cuda.profiler.start()

    def stop_profile(self):
        torch.cuda.profiler.stop()

    # ------------------------------------------------------------------ TP control plane (:524-538, :722-755)
    def start_worker_execution_loop(self) -> None:
        """Non-driver ranks: mirror the driver's calls until it sends the empty record (execute_model(None))."""
        assert self._rank != self._driver_rank, "the driver rank calls execute_model, not the loop"
        while self._run_non_driver_rank():
            pass

    def _run_non_driver_rank(self) -> bool:
        """One mirrored step; False when the driver signalled the end of the loop (:722-755)."""
        assert self._rank != self._driver_rank
        data = self._tp.broadcast_object(None, src=self._driver_rank) if self._tp is not None else {}
        if not data:
            return False
        self._execute(data["request"], data["no_spec"], data["disable_all_speculation"], data["num_lookahead_slots"])
        return True

    # ------------------------------------------------------------------ execute_model (:461-560)
    @torch.no_grad()   # (the reference uses inference_mode; tensors made under it could not be updated by a later
    # hipGraph capture outside it -- the CUDA generator's graph-safe state among them -- and nothing here needs it)
    def execute_model(self, execute_model_req: Optional[ExecuteModelRequest] = None) -> List[SamplerOutput]:
        if self._rank != self._driver_rank:
            self._run_non_driver_rank()
            return []
        if execute_model_req is None:
            # the signal that ends start_worker_execution_loop() on the other ranks (:470-478)
            if self._tp is not None:
                self._tp.broadcast_object({}, src=self._driver_rank)
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
        if self._tp is not None:   # broadcast_tensor_dict of the control scalars (+ the inputs, worker_base.py:340-344)
            self._tp.broadcast_object(dict(num_lookahead_slots=num_lookahead_slots, no_spec=no_spec,
                                           disable_all_speculation=disable_all_speculation,
                                           run_spec_proposer_for_prefill=any(s.is_prompt for s in sgml),
                                           request=execute_model_req), src=self._driver_rank)
        return self._execute(execute_model_req, no_spec, disable_all_speculation, num_lookahead_slots)

    def _execute(self, req: ExecuteModelRequest, no_spec: bool, disable_all_speculation: bool,
                 num_lookahead_slots: int) -> List[SamplerOutput]:
        self._track_finished_requests(req)
        if not req.seq_group_metadata_list:
            return []
        if no_spec:
            return self._run_no_spec(req, skip_proposer=True)      # skip_proposer forced (:699)
        return self._run_speculative_decoding_step(req, num_lookahead_slots)

    def _should_disable_all_speculation(self, req: ExecuteModelRequest) -> bool:
        return self.disable_by_batch_size is not None and req.running_queue_size >= self.disable_by_batch_size

    def _track_finished_requests(self, req: ExecuteModelRequest) -> None:
        """:1178-1188 plus the slot bookkeeping: a finished request gives its engine slot back."""
        for rid in req.finished_requests_ids:
            for seq_id in self._request_id_seq_id_mapping.get(rid, ()):
                self._seq_with_bonus_token_in_last_step.discard(seq_id)
            self._request_id_seq_id_mapping.pop(rid, None)
            slot = self._slots.pop(rid, None)
            if slot is not None:
                self.engine.free_slot(slot)

    @staticmethod
    def _only_seq(s: SequenceGroupMetadata) -> Tuple[int, SequenceData]:
        if len(s.seq_data) != 1:
            raise NotImplementedError("beam / parallel sampling groups are not on the QSpec path (one sequence per request)")
        return next(iter(s.seq_data.items()))

    def _admit(self, prompts: List[SequenceGroupMetadata]) -> List[int]:
        """Prompts (or preempted requests coming back for recomputation) take the first free slots; their prompt pass
        is ONE varlen forward (engine.add_sequences_to).  The engine validates every slot, prompt and block table
        before its first forward; should the admission still fail half-way (several prompt passes, one raising), the
        slots it had occupied are given back, so that worker and engine never disagree about which slots are free."""
        for s in prompts:
            if s.request_id in self._slots:          # recomputation: start over
                self.engine.free_slot(self._slots.pop(s.request_id))
        free = [b for b in range(self.max_num_seqs) if b not in self._slots.values()]
        if len(free) < len(prompts):
            raise RuntimeError(f"no free sequence slot for request {prompts[len(free)].request_id}: "
                               f"max_num_seqs={self.max_num_seqs} are running")
        slots, toks, tables = free[:len(prompts)], [], []
        for s, b in zip(prompts, slots):
            seq_id, data = self._only_seq(s)
            toks.append(data.get_token_ids())
            tables.append(s.block_tables.get(seq_id) if s.block_tables else None)
            sp = s.sampling_params   # before the prompt pass: the first token is sampled with them too
            self.engine.set_sampling_params(b, getattr(sp, "temperature", 0.0) or 0.0, getattr(sp, "top_k", -1) or -1,
                                            getattr(sp, "top_p", 1.0) if sp is not None else 1.0)
        try:
            self.engine.add_sequences_to(slots, toks, tables)
        except Exception:
            for b in slots:
                self.engine.free_slot(b)
            raise
        for s, slot in zip(prompts, slots):
            self._slots[s.request_id] = slot
            self._request_id_seq_id_mapping[s.request_id].add(self._only_seq(s)[0])
        return slots

    def _decode_slots(self, sgml) -> List[int]:
        """Engine slots of a decode batch in request order; refreshes block tables (the engine uploads a table only when
        it changed); cross-checks the lengths."""
        slots = []
        for s in sgml:
            if s.request_id not in self._slots:
                raise KeyError(f"request {s.request_id} was never prefilled on this worker (or has finished)")
            slot = self._slots[s.request_id]
            seq_id, data = self._only_seq(s)
            if s.block_tables and s.block_tables.get(seq_id) is not None:
                self.engine.set_block_table(slot, s.block_tables[seq_id])
            if data.output_token_ids and data.get_len() != self.engine._len_ub[slot]:
                raise ValueError(f"request {s.request_id}: the scheduler holds {data.get_len()} tokens, the worker "
                                 f"{self.engine._len_ub[slot]}: outputs were dropped or re-ordered between steps")
            slots.append(slot)
        if len(set(slots)) != len(slots):
            raise ValueError("a request appears twice in seq_group_metadata_list")
        return slots

    def _read_cycle(self):
        """The one host read of a cycle: output tokens + the cycle's error word.  On a timed-out device-side hand-off the
        cycle is re-run once without hand-offs (engine.recover(), bit-identical results); every rank reads the same
        all-reduced word, so all ranks re-run -- or raise -- together."""
        out, err = self.engine.read_outputs()
        if err:
            self.engine.recover()
            out, err = self.engine.read_outputs()
            if err:
                raise DeviceHandoffTimeout("a device-side hand-off timed out during this cycle and again in the replay "
                                           "without hand-offs (results invalid)")
        return out

    def _run_no_spec(self, req: ExecuteModelRequest, skip_proposer: bool) -> List[SamplerOutput]:
        """:666-720.  Prompts: W4A16 prefill, first token sampled by the target.  Decode sequences with speculation off
        for this step: the scorer alone, one token each.  A batch may hold both (prompts first, as vLLM orders them).
        Output: `_serialize_sampler_output_no_logprobs` (:582-664)."""
        sgml = req.seq_group_metadata_list
        assert req.w4a4 is False
        tokens = [VLLM_INVALID_TOKEN_ID] * len(sgml)
        prompts = [i for i, s in enumerate(sgml) if s.is_prompt]
        decodes = [i for i, s in enumerate(sgml) if not s.is_prompt]
        dslots = self._decode_slots([sgml[i] for i in decodes])     # before admissions: validates the running ones
        if prompts:
            slots = self._admit([sgml[i] for i in prompts])
            first = self.engine.gen_tokens[:, 0].cpu()
            self._check_out_of_cycle_errors("the prompt pass")
            for i, slot in zip(prompts, slots):
                tokens[i] = int(first[slot])
            self.scorer_calls += 1          # one scorer call for the prompt batch, as in the reference
        if decodes:
            tok = self._no_spec_step(dslots)
            for i, slot in zip(decodes, dslots):
                tokens[i] = tok[slot]
        return [self._serialize_sampler_output_no_logprobs(sgml, tokens)]

    def _check_out_of_cycle_errors(self, what: str) -> None:
        """Prompt passes and non-speculative steps run outside the captured cycle (other stream, no error-word collection
        at their end): their launches are followed by a host read anyway, so the sticky words of EVERY hand-off workspace
        of the device and of the one-shot all-reduce are checked from the host there.  No replay for these: the call fails."""
        if self.engine.error_flag():
            raise DeviceHandoffTimeout(f"a device-side hand-off timed out during {what} (results invalid)")

    def _no_spec_step(self, dslots: List[int]) -> Dict[int, int]:
        self.engine.step_no_spec(participants=dslots)
        self.scorer_calls += 1
        out = self.engine.out_tokens[:, 0].cpu()
        self._check_out_of_cycle_errors("a non-speculative decode step")
        self.engine.note_emitted([1 if b in dslots else 0 for b in range(self.max_num_seqs)])
        return {b: int(out[b]) for b in dslots}

    def _serialize_sampler_output_no_logprobs(self, sgml, tokens: List[int]) -> SamplerOutput:
        """:582-664: only the token ids are populated; a sequence group without a sample (`do_sample` False: a
        non-terminal prompt chunk) still gets its own, empty, output."""
        outs = []
        for s, tok in zip(sgml, tokens):
            if not s.do_sample:
                outs.append(CompletionSequenceGroupOutput(samples=[], prompt_logprobs=None))
                continue
            seq_id, data = self._only_seq(s)
            sp = s.sampling_params
            prompt_logprobs = None
            if s.is_prompt and sp is not None and sp.prompt_logprobs is not None and sp.prompt_logprobs > 0:
                prompt_logprobs = [create_logprobs_output(t, -1, 0.0, [], []) for t in data.get_prompt_token_ids()[1:]]
            outs.append(create_sequence_group_output(tok, -1, 0.0, seq_id, [], [], prompt_logprobs))
        return SamplerOutput(outputs=outs)

    def _run_speculative_decoding_step(self, req: ExecuteModelRequest, num_lookahead_slots: int) -> List[SamplerOutput]:
        """:758-858 as one graph replay: proposals (w4a4=True) -> scoring (w4a4=False) -> verification.  Requests with
        `num_speculative_tokens == 0` in an otherwise speculative batch get the target's one token (the reference scores
        them with proposal length 0, top1_proposer.py:103-135): they sit out the cycle and take the scorer-only step."""
        k = self.engine.k
        if num_lookahead_slots != k:
            raise ValueError(f"num_lookahead_slots = {num_lookahead_slots}, but the cycle was captured for "
                             f"num_speculative_tokens = {k} (SpeculativeConfig.num_speculative_tokens)")
        sgml = req.seq_group_metadata_list
        if any(s.is_prompt for s in sgml):
            raise NotImplementedError("prompt chunks inside a speculative step (chunked prefill) are not scheduled onto "
                                      "this worker: vLLM sends prompt-only batches with num_lookahead_slots == 0")
        slots = self._decode_slots(sgml)
        spec = [i for i, s in enumerate(sgml) if s.num_speculative_tokens != 0]
        nospec = [i for i, s in enumerate(sgml) if s.num_speculative_tokens == 0]
        req.w4a4 = True                                   # :799
        self.proposer_calls += k
        scorer_req = req.clone(sgml)                      # mqa_scorer.py:65: clone drops w4a4 -> W4A16
        assert scorer_req.w4a4 is False
        req.w4a4 = False                                  # :812
        self.scorer_calls += 1
        spec_slots = [slots[i] for i in spec]
        self.engine.step(participants=spec_slots)
        out = self._read_cycle()
        self.engine.note_emitted([int((out[b] != -1).sum()) if b in spec_slots else 0 for b in range(out.shape[0])])
        rows = [[VLLM_INVALID_TOKEN_ID] * (k + 1) for _ in sgml]
        for i in spec:
            rows[i] = out[slots[i]].tolist()
        if nospec:
            tok = self._no_spec_step([slots[i] for i in nospec])
            for i in nospec:
                rows[i][0] = tok[slots[i]]
        return self._create_output_sampler_list(sgml, rows, k, slots)

    def _create_output_sampler_list(self, sgml, rows: List[List[int]], k: int, slots: List[int]) -> List[SamplerOutput]:
        """:972-1063: [batch, k+1] accepted token ids -> one SamplerOutput per step, padded with -1 so that every
        sequence has the same number of outputs; the list stops at the first step in which no sequence emitted a token
        (:1023-1026); metrics ride on the first one."""
        batch_size, num_steps = len(sgml), k + 1
        by_step = [[rows[b][j] for b in range(batch_size)] for j in range(num_steps)]
        if self._disable_logprobs:
            ranks, lps, topk_lps, topk_ids = self._create_dummy_logprob_lists(batch_size, num_steps, self.max_logprobs)
        else:
            ranks, lps, topk_lps, topk_ids = self._create_logprob_lists_from_tensors(by_step, slots, self.max_logprobs)
        seq_ids = [self._only_seq(s)[0] for s in sgml]                                   # get_all_seq_ids_and_request_ids
        num_logprobs_per_seq = [(s.sampling_params.logprobs or 0) if s.sampling_params is not None else 0
                                for s in sgml]                                           # get_all_num_logprobs
        outs: List[SamplerOutput] = []
        for step in range(num_steps):
            if all(t == VLLM_INVALID_TOKEN_ID for t in by_step[step]):
                break
            outs.append(SamplerOutput(outputs=[
                create_sequence_group_output(
                    token_id=by_step[step][b], token_id_logprob_rank=ranks[step][b], token_id_logprob=lps[step][b],
                    seq_id=seq_ids[b], topk_token_ids=topk_ids[step][b][:num_logprobs_per_seq[b]],
                    topk_logprobs=topk_lps[step][b][:num_logprobs_per_seq[b]])
                for b in range(batch_size)]))
        self._track_sequences_with_bonus_tokens(sgml, by_step)
        if outs and not self._disable_log_stats:
            m = self._metrics.maybe_collect_rejsample_metrics(k)
            if m is not None:
                outs[0].spec_decode_worker_metrics = m
        return outs

    @staticmethod
    def _create_dummy_logprob_lists(batch_size: int, num_steps: int, num_top_k: int):
        """:1083-1127: ranks -1, logprobs 0.0, top-k entries None (dropped by create_logprobs_output)."""
        ranks = [[-1] * batch_size for _ in range(num_steps)]
        lps = [[0.0] * batch_size for _ in range(num_steps)]
        topk_lps = [[[None] * num_top_k for _ in range(batch_size)] for _ in range(num_steps)]
        topk_ids = [[[None] * num_top_k for _ in range(batch_size)] for _ in range(num_steps)]
        return ranks, lps, topk_lps, topk_ids

    def _create_logprob_lists_from_tensors(self, by_step: List[List[int]], slots: List[int], num_top_k: int):
        """:1129-1176 (`disable_logprobs=False`): log of the target's probabilities of the cycle, the accepted token's
        logprob and rank (vllm/spec_decode/util.py:34-52) and the top-k per position.  Host-side bookkeeping with torch
        on the engine's [B, k+1, V] target distribution -- not on the default path, not on the metric's path."""
        probs = self.engine.target_probs[slots]                               # [batch, k+1, V]
        logp = torch.log(probs.clamp_min(torch.finfo(torch.float32).tiny)).transpose(0, 1)   # [k+1, batch, V]
        ids = torch.tensor(by_step, dtype=torch.int64, device=logp.device)    # [k+1, batch], -1 = nothing
        safe = ids.clamp_min(0)
        sel = logp.gather(-1, safe.unsqueeze(-1)).squeeze(-1)
        ranks = (logp > sel.unsqueeze(-1)).sum(-1) + 1
        top_lp, top_id = logp.topk(min(num_top_k, logp.shape[-1]), dim=-1)
        return ranks.tolist(), sel.tolist(), top_lp.tolist(), top_id.tolist()

    def _track_sequences_with_bonus_tokens(self, sgml, accepted_token_ids_by_step: List[List[int]]) -> None:
        """:1190-1210: a sequence whose last step position holds a token (!= -1) received the bonus token."""
        for idx, s in enumerate(sgml):
            seq_id, _ = self._only_seq(s)
            if accepted_token_ids_by_step[-1][idx] == VLLM_INVALID_TOKEN_ID:
                self._seq_with_bonus_token_in_last_step.discard(seq_id)
            else:
                self._seq_with_bonus_token_in_last_step.add(seq_id)
            self._request_id_seq_id_mapping[s.request_id].add(seq_id)
