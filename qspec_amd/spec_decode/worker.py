"""Worker boundary: what vLLM's executor calls.

`create_spec_worker(**kwargs) -> SpecDecodeWorker` and the methods vLLM's WorkerBase contract needs
(`init_device`, `load_model`, `determine_num_available_blocks`, `initialize_cache`, `execute_model`,
`get_cache_block_size_bytes`, `start_worker_execution_loop`, `rank`, `device`), mirroring
vllm/spec_decode/spec_decode_worker.py:53-113,118-470,461-560,722-755,972-1063,1178-1210.  vLLM itself is not
importable here, so the request / output records are small dataclasses with the reference's field names
(vllm/sequence.py).

QSpec specifics preserved:
  * proposer and scorer are the SAME model object and the SAME KV cache (:339-345, :421-444);
  * `execute_model_req.w4a4 = True` only around the proposer (:797-812); `ExecuteModelRequest.clone()` does not
    carry `w4a4`, so the scorer always runs W4A16 (vllm/sequence.py:1301,1331-1348);
  * prefill never runs the proposer (:699);
  * one `SamplerOutput` per emitted position, `-1` = no token for that sequence (:972-1063);
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

from collections import defaultdict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Set, Tuple

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

    def get_token_ids(self):
        return list(self.prompt_token_ids) + list(self.output_token_ids)


@dataclass
class SequenceGroupMetadata:
    request_id: str
    is_prompt: bool
    seq_data: Dict[int, SequenceData]
    num_speculative_tokens: Optional[int] = None
    block_tables: Optional[Dict[int, List[int]]] = None   # seq_id -> physical block ids (vllm/sequence.py)


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
class SamplerOutput:
    """One decode position for the whole batch (vllm/model_executor/layers/sampler.py SamplerOutput)."""
    sampled_token_ids: torch.Tensor                      # [n] int64 on the host, request order, -1 = nothing emitted
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
    pipeline_parallel_size, rank."""
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
                            disable_log_stats=kwargs.get("disable_log_stats", False),
                            engine_factory=kwargs.get("engine_factory"))


class SpecDecodeWorker:
    def __init__(self, model_config: QuarotLlamaConfig, speculative_config: SpeculativeConfig, max_num_seqs: int,
                 max_model_len: int, block_size: int, device: str, model: Optional[QuarotLlamaForCausalLM] = None,
                 seed: int = 0, rank: int = 0, disable_log_stats: bool = False, engine_factory=None):
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
        self._engine_factory = engine_factory or QSpecEngine
        self.engine: Optional[QSpecEngine] = None
        self._metrics: Optional[AsyncMetricsCollector] = None
        self._slots: Dict[str, int] = {}                 # request_id -> engine slot
        # :1178-1210 -- sequences that received a bonus token in their last step / request -> its sequence ids
        self._seq_with_bonus_token_in_last_step: Set[int] = set()
        self._request_id_seq_id_mapping: Dict[str, Set[int]] = defaultdict(set)
        self.proposer_calls = 0     # forwards run with w4a4=True (for tests of the toggle)
        self.scorer_calls = 0

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
        """:326-369: the scorer loads the model, the proposer receives the very same object."""
        if self._device.type == "cuda":
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
        self.engine = self._engine_factory(self._model, self.speculative_config.num_speculative_tokens,
                                           self.max_num_seqs, self.max_model_len, self.block_size, seed=self._seed)
        self._metrics = AsyncMetricsCollector(self.engine.sampler)
        self._metrics.init_gpu_tensors(self._rank)

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
    def _request_ids(sgml):
        return [s.request_id for s in sgml]

    @staticmethod
    def _only_seq(s: SequenceGroupMetadata) -> Tuple[int, SequenceData]:
        if len(s.seq_data) != 1:
            raise NotImplementedError("beam / parallel sampling groups are not on the QSpec path (one sequence per request)")
        return next(iter(s.seq_data.items()))

    def _admit(self, prompts: List[SequenceGroupMetadata]) -> List[int]:
        """Prompts (or preempted requests coming back for recomputation) take the first free slots; their prompt pass
        is ONE varlen forward (engine.add_sequences_to)."""
        for s in prompts:
            if s.request_id in self._slots:          # recomputation: start over
                self.engine.free_slot(self._slots.pop(s.request_id))
        free = [b for b in range(self.max_num_seqs) if b not in self._slots.values()]
        if len(free) < len(prompts):
            raise RuntimeError(f"no free sequence slot for request {prompts[len(free)].request_id}: "
                               f"max_num_seqs={self.max_num_seqs} are running")
        slots, toks, tables = free[:len(prompts)], [], []
        for s in prompts:
            seq_id, data = self._only_seq(s)
            toks.append(data.get_token_ids())
            tables.append(s.block_tables.get(seq_id) if s.block_tables else None)
        self.engine.add_sequences_to(slots, toks, tables)
        for s, slot in zip(prompts, slots):
            self._slots[s.request_id] = slot
            self._request_id_seq_id_mapping[s.request_id].add(self._only_seq(s)[0])
        return slots

    def _decode_slots(self, sgml) -> List[int]:
        """Engine slots of a decode batch in request order; refreshes block tables; cross-checks the lengths."""
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

    def _run_no_spec(self, req: ExecuteModelRequest, skip_proposer: bool) -> List[SamplerOutput]:
        """:666-720.  Prompts: W4A16 prefill, first token sampled by the target.  Decode sequences with speculation off
        for this step: the scorer alone, one token each.  A batch may hold both (prompts first, as vLLM orders them)."""
        sgml = req.seq_group_metadata_list
        assert req.w4a4 is False
        ids = self._request_ids(sgml)
        tokens = torch.full((len(sgml),), -1, dtype=torch.int64)
        prompts = [i for i, s in enumerate(sgml) if s.is_prompt]
        decodes = [i for i, s in enumerate(sgml) if not s.is_prompt]
        dslots = self._decode_slots([sgml[i] for i in decodes])     # before admissions: validates the running ones
        if prompts:
            slots = self._admit([sgml[i] for i in prompts])
            first = self.engine.gen_tokens[:, 0].cpu()
            for i, slot in zip(prompts, slots):
                tokens[i] = first[slot]
            self.scorer_calls += 1          # one scorer call for the prompt batch, as in the reference
        if decodes:
            self.engine.step_no_spec(participants=dslots)
            self.scorer_calls += 1
            out = self.engine.out_tokens[:, 0].cpu()
            for i, slot in zip(decodes, dslots):
                tokens[i] = out[slot]
            self.engine.note_emitted([1 if b in dslots else 0 for b in range(self.max_num_seqs)])
        return [SamplerOutput(tokens, ids)]

    def _run_speculative_decoding_step(self, req: ExecuteModelRequest, num_lookahead_slots: int) -> List[SamplerOutput]:
        """:758-858 as one graph replay: proposals (w4a4=True) -> scoring (w4a4=False) -> verification."""
        k = self.engine.k
        assert num_lookahead_slots == k, "the cycle graph is captured for a fixed k"
        sgml = req.seq_group_metadata_list
        if any(s.is_prompt for s in sgml):
            raise NotImplementedError("prompt chunks inside a speculative step (chunked prefill) are not scheduled onto "
                                      "this worker: vLLM sends prompt-only batches with num_lookahead_slots == 0")
        slots = self._decode_slots(sgml)
        req.w4a4 = True                                   # :799
        self.proposer_calls += k
        scorer_req = req.clone(sgml)                      # mqa_scorer.py:65: clone drops w4a4 -> W4A16
        assert scorer_req.w4a4 is False
        req.w4a4 = False                                  # :812
        self.scorer_calls += 1
        self.engine.step(participants=slots)
        return self._create_output_sampler_list(sgml, slots, k)

    def _create_output_sampler_list(self, sgml, slots: List[int], k: int) -> List[SamplerOutput]:
        """:972-1063: transpose [B, k+1] -> k+1 per-step outputs; metrics ride on the first one."""
        out = self.engine.out_tokens.cpu()                # the one host sync of the cycle (reference: three)
        if self.engine.error_flag():
            raise RuntimeError("a device-side hand-off timed out during this cycle (results invalid)")
        self.engine.note_emitted([int((out[b] != -1).sum()) if b in slots else 0 for b in range(out.shape[0])])
        ids = self._request_ids(sgml)
        by_step = [out[slots, j].clone() for j in range(k + 1)]
        self._track_sequences_with_bonus_tokens(sgml, by_step)
        outs = [SamplerOutput(t, ids) for t in by_step]
        # drop trailing steps in which no sequence emitted anything (:1038-1046)
        while len(outs) > 1 and bool((outs[-1].sampled_token_ids == -1).all()):
            outs.pop()
        if not self._disable_log_stats:
            outs[0].spec_decode_worker_metrics = self._metrics.maybe_collect_rejsample_metrics(k)
        return outs

    def _track_sequences_with_bonus_tokens(self, sgml, accepted_token_ids_by_step) -> None:
        """:1190-1210: a sequence whose last step position holds a token (!= -1) received the bonus token."""
        for idx, s in enumerate(sgml):
            seq_id, _ = self._only_seq(s)
            if int(accepted_token_ids_by_step[-1][idx]) == -1:
                self._seq_with_bonus_token_in_last_step.discard(seq_id)
            else:
                self._seq_with_bonus_token_in_last_step.add(seq_id)
            self._request_id_seq_id_mapping[s.request_id].add(seq_id)
