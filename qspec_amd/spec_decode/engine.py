"""QSpecEngine: the draft -> verify -> accept cycle with all state resident on the GPU.

What the reference does per step on the host (vllm/spec_decode/spec_decode_worker.py:758-1063) --
toggle `w4a4`, run the k-step draft loop (draft_model_runner.py:169-376), build the MQA scoring batch
(mqa_scorer.py:12-114), run the target, rejection-sample, `.tolist()` three times -- becomes here ONE
captured hipGraph per cycle:

    prepare_draft -> k x [ forward(W4A4) -> lm_head -> softmax/argmax -> advance_step ]
                  -> prepare_verify -> forward(W4A16) -> lm_head -> softmax/argmax
                  -> rejection sample -> commit

The draft and the verify pass share one packed-int4 weight buffer per layer and one paged KV cache
(spec_decode_worker.py:339-345,421-444; vllm/worker/worker.py:309-327); the verify pass rewrites the
k+1 KV slots the draft pass wrote (mqa_scorer.py:42-60), which is also why no bonus-token batch
expansion is needed (multi_step_worker.py:74-80).

Batch membership: the graph is captured for `max_batch` SLOTS; a slot is empty while seq_lens[slot] == 0 and then
runs through the cycle as a dummy that writes nothing and emits nothing (include/qspec_hip.h, spec-decode glue), so
requests join (`add_sequence`) and leave (`free_slot`) between replays without re-capturing.  Block tables are per
slot and may be set from the scheduler's tables (`set_block_table`); by default slot b owns a contiguous range.

Capacity is part of the contract: `step()` refuses (ValueError, nothing enqueued) a cycle that could write beyond a
sequence's block table / max_model_len / output buffer, from a host-side upper bound of every sequence length that is
exact after `sync_lens()` (the worker syncs once per cycle anyway) and grows by k+1 per unsynced step otherwise.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import os

import torch

from .. import ops
from ..model import AttentionMetadata, QuarotLlamaForCausalLM, Scratch
from .metrics import SpecDecodeWorkerMetrics, metrics_from_counters
from .rejection_sampler import RejectionSampler

ATTN_CHUNK = 128  # keys per context split of the attention kernel (QS_ATT_CHUNK)


def n_splits_for(ctx: int, n_groups: int = 0) -> int:
    """Context splits of the attention kernel.  Decode (n_groups = sequences x kv heads x row blocks > 0): enough
    workgroups to occupy the 256 CUs (8 at batch 4 -- measured best: 6 and 12+ are slower), independent of the context
    length: a split longer than 128 keys is walked in 128-key chunks inside the kernel, so the partials the next launch
    merges do not grow with the context.  Prefill-sized queries (n_groups = 0) fill the chip with row blocks: 1."""
    if n_groups > 0:
        forced = int(os.environ.get("QSPEC_ATTN_SPLITS", "0"))   # dev knob for sweeps
        if forced > 0:
            return forced
        n = max(1, min(16, (256 + n_groups - 1) // n_groups))
        if n_groups > 64 and 256 % n_groups:
            # group counts that do not divide the chip (Llama-2-13B at batch 4: 4 x 40 kv heads = 160): two workgroups are
            # resident per CU, so fill the 512 slots -- 160 x 3 = 480 workgroups of 171 keys instead of 320 of 256, of which
            # 64 CUs carried two (cycle 13.49 -> 12.99 ms; 4 splits = 640 workgroups: 13.28)
            n = max(n, min(16, 512 // n_groups))
        return n
    return 1


class QSpecEngine:
    def __init__(self, model: QuarotLlamaForCausalLM, num_speculative_tokens: int = 3, max_batch: int = 4,
                 max_model_len: int = 1024, block_size: int = 16, max_new_tokens: int = 1024, use_graph: bool = True,
                 seed: int = 0, num_blocks: Optional[int] = None, acceptance_sampler=None):
        self.model = model
        self.cfg = cfg = model.config
        if max_model_len > cfg.max_position_embeddings:
            # the RoPE table has max_position_embeddings rows (quarot_llama.py:112-120); vLLM refuses such a
            # max_model_len at config time -- a position beyond it would read past the table
            raise ValueError(f"max_model_len {max_model_len} > max_position_embeddings {cfg.max_position_embeddings}")
        self.k = k = num_speculative_tokens
        self.B = B = max_batch
        self.block_size = block_size
        self.max_model_len = max_model_len
        self.use_graph = use_graph
        dev = self.device = model.device
        i64, i32 = torch.int64, torch.int32
        # ---- one KV cache for both passes: [num_blocks, block_size, n_kv, d] per layer
        # num_blocks: what the worker's initialize_cache(num_gpu_blocks) hands over (the scheduler allocates block ids
        # below it and sends block tables); default: every slot owns a contiguous range of max_model_len tokens
        self.blocks_per_seq = (max_model_len + block_size - 1) // block_size
        self.num_blocks = B * self.blocks_per_seq if num_blocks is None else int(num_blocks)
        if self.num_blocks < 1:
            raise ValueError("the KV cache needs at least one block")
        shape = (self.num_blocks, block_size, cfg.num_key_value_heads, cfg.head_dim)
        self.kv_caches = [(torch.zeros(shape, dtype=torch.float16, device=dev),
                           torch.zeros(shape, dtype=torch.float16, device=dev)) for _ in range(cfg.num_hidden_layers)]
        if self.num_blocks >= B * self.blocks_per_seq:
            self.block_tables = torch.arange(B * self.blocks_per_seq, dtype=i32, device=dev).view(B, self.blocks_per_seq).contiguous()
            self._capacity = [self.blocks_per_seq * block_size] * B    # tokens the slot's block table covers (host)
        else:   # fewer blocks than max_num_seqs x max_model_len (vLLM's normal case): every request brings its table
            self.block_tables = torch.zeros(B, self.blocks_per_seq, dtype=i32, device=dev)
            self._capacity = [0] * B
        self._bt_host: List[Optional[List[int]]] = [None] * B      # the table last uploaded per slot (None: the default)
        self._len_ub = [0] * B                                     # host upper bound of seq_lens (0 = empty slot)
        self._gen_ub = [0] * B                                     # host upper bound of gen_lens
        # ---- sequence state (seq_lens[b] == 0: empty slot)
        self.seq_lens = torch.zeros(B, dtype=i32, device=dev)      # L: tokens known (KV valid below L-1)
        self.last_token = torch.zeros(B, dtype=i64, device=dev)
        self.gen_tokens = torch.full((B, max_new_tokens + k + 1), -1, dtype=i64, device=dev)
        self.gen_lens = torch.zeros(B, dtype=i32, device=dev)
        self.n_active = 0
        # sequences taking part in the next step (the scheduler may leave a running request out of a step): the cycle
        # works on eff_lens = seq_lens * step_mask, so a request that sits out looks like an empty slot and keeps its state
        self.step_mask = torch.ones(B, dtype=i32, device=dev)
        self.eff_lens = torch.zeros(B, dtype=i32, device=dev)
        self._mask_host = [1] * B
        self._len_before, self._gen_before = [0] * B, [0] * B
        # ---- per-cycle buffers
        V = cfg.vocab_size
        n_splits = n_splits_for(max_model_len, B * cfg.num_key_value_heads)
        self.d_tokens = torch.zeros(B, dtype=i64, device=dev)
        self.d_pos = torch.zeros(B, dtype=i64, device=dev)
        self.d_slots = torch.zeros(B, dtype=i64, device=dev)
        self.d_ctx = torch.zeros(B, dtype=i32, device=dev)
        self.d_qstart = torch.arange(B + 1, dtype=i32, device=dev)
        self.draft_probs_kbv = torch.zeros(k, B, V, dtype=torch.float32, device=dev)   # step-major
        self.draft_ids_kb = torch.zeros(k, B, dtype=i64, device=dev)
        T = B * (k + 1)
        self.v_tokens = torch.zeros(T, dtype=i64, device=dev)
        self.v_pos = torch.zeros(T, dtype=i64, device=dev)
        self.v_slots = torch.zeros(T, dtype=i64, device=dev)
        self.v_ctx = torch.zeros(B, dtype=i32, device=dev)
        self.v_qstart = (torch.arange(B + 1, dtype=i32, device=dev) * (k + 1)).contiguous()
        self.target_probs = torch.zeros(B, k + 1, V, dtype=torch.float32, device=dev)
        self.target_tokens = torch.zeros(B, k + 1, dtype=i64, device=dev)
        # the cycle's output and its error word in ONE buffer: the worker's single host read per cycle
        self._out_err = torch.full((B * (k + 1) + 1,), -1, dtype=i64, device=dev)
        self._out_err[-1] = 0
        self.out_tokens = self._out_err[:B * (k + 1)].view(B, k + 1)
        self.err_word = self._out_err[B * (k + 1):]               # OR of the sticky device-side error words (0 = fine)
        self._out_err_host = torch.empty(B * (k + 1) + 1, dtype=i64, pin_memory=(dev.type == "cuda"))
        self._snap_i32 = torch.zeros(2 * B, dtype=i32, device=dev)  # sequence state at the start of the last cycle
        self._snap_i64 = torch.zeros(B + 5, dtype=i64, device=dev)
        self.recoveries = 0                                         # cycles re-run without device-side hand-offs
        self.accepted = torch.zeros(B, k, dtype=torch.uint8, device=dev)
        self.recovered = torch.zeros(B, k, dtype=i64, device=dev)
        # the acceptance sampler (spec_decode_worker.py:95-110): RejectionSampler by default, or the instance handed in
        # (TypicalAcceptanceSampler) -- same call surface, same counters
        self.sampler = acceptance_sampler if acceptance_sampler is not None else RejectionSampler(seed=seed)
        self.sampler.init_gpu_tensors(str(dev))
        self.scratch_draft = Scratch(cfg, B, B, 1, n_splits, dev)
        self.scratch_verify = Scratch(cfg, T, B, k + 1, n_splits, dev)
        self.md_draft = AttentionMetadata(self.d_slots, self.block_tables, self.d_ctx, self.d_qstart, 1, n_splits)
        self.md_verify = AttentionMetadata(self.v_slots, self.block_tables, self.v_ctx, self.v_qstart, k + 1, n_splits)
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._graph_draft: Optional[torch.cuda.CUDAGraph] = None   # fallback: proposer only (verify pass eager)
        # ---- sampling parameters per slot (Sampler.forward's SamplingTensors, sampler.py:216-316).  temperature 0 = greedy
        # (vLLM resets top_k / top_p of such requests itself, sampling_params.py).  While every occupied slot is greedy the
        # cycle is the greedy one above (fused lm_head + softmax); as soon as one is not, the cycle samples EVERY row through
        # ops.sample_top_k_top_p (greedy rows take the argmax there) -- a second captured graph, same state, same RNG stream
        self.samp_temp = torch.zeros(B, dtype=torch.float32, device=dev)
        self.samp_topk = torch.full((B,), -1, dtype=torch.int32, device=dev)
        self.samp_topp = torch.ones(B, dtype=torch.float32, device=dev)
        self.v_samp_temp = torch.zeros(T, dtype=torch.float32, device=dev)     # the same, one row per verify token
        self.v_samp_topk = torch.full((T,), -1, dtype=torch.int32, device=dev)
        self.v_samp_topp = torch.ones(T, dtype=torch.float32, device=dev)
        self._samp_host = [(0.0, -1, 1.0)] * B
        self._mode_sampling = False
        self._graph_s: Optional[torch.cuda.CUDAGraph] = None       # the cycle with the sampling front end
        self._graph_draft_s: Optional[torch.cuda.CUDAGraph] = None
        # test hooks: injected random draws for the rejection sampler (eager mode only)
        self.inject_uniform: Optional[torch.Tensor] = None
        self.inject_exponential: Optional[torch.Tensor] = None
        self._prefill_scratch: Optional[Scratch] = None

    # the embedding lookups of the cycle's four forwards ride in the bookkeeping launches in front of them (5 launches less)
    EMBED_IN_BOOKKEEPING = os.environ.get("QSPEC_EMBED_IN_BOOKKEEPING", "1") != "0"
    PREFILL_TOKEN_BUDGET = 4096   # tokens per prompt-pass forward when several requests are admitted together

    # ------------------------------------------------------------------ prefill (_run_no_spec, :666-720)
    @torch.no_grad()
    def add_sequences(self, prompts: Sequence[Sequence[int]]):
        """Prompt pass for slots 0 .. len(prompts)-1 (a full batch in the benchmarks and most tests): ONE varlen
        forward over all prompts (vLLM batches prompts the same way), the target's greedy token per prompt."""
        assert len(prompts) <= self.B
        self.add_sequences_to(list(range(len(prompts))), prompts)

    @torch.no_grad()
    def add_sequences_to(self, slots: Sequence[int], prompts: Sequence[Sequence[int]],
                         block_tables: Optional[Sequence[Optional[Sequence[int]]]] = None) -> None:
        """Admit several requests at once: one W4A16 forward over the concatenated prompts (flash-attn varlen form)."""
        cfg, dev = self.cfg, self.device
        n = len(slots)
        if n == 0:
            return
        # Validate EVERY slot, prompt and block table before the first forward: a later request that does not fit must
        # not leave earlier ones of the same call admitted (their slots would be occupied here and free in the worker).
        if len(set(slots)) != n:
            raise ValueError("a slot appears twice in one admission")
        for i, b in enumerate(slots):
            if not 0 <= b < self.B or self._len_ub[b] != 0:
                raise ValueError(f"slot {b} is not an empty slot of this engine")
            table = None if block_tables is None else block_tables[i]
            if table is not None:
                self._validate_block_table(table)
            cap = self._capacity[b] if table is None else len(table) * self.block_size
            T = len(prompts[i])
            if T < 1 or T + 1 > min(cap, self.max_model_len):
                raise ValueError(f"prompt of {T} tokens does not fit slot {b} (capacity {cap} tokens, max_model_len "
                                 f"{self.max_model_len})" + ("" if cap else ": the request brought no block table"))
        if n == 1:
            self.add_sequence(slots[0], prompts[0], None if block_tables is None else block_tables[0])
            return
        if sum(len(p) for p in prompts) > self.PREFILL_TOKEN_BUDGET:
            # several passes of at most PREFILL_TOKEN_BUDGET tokens (measured: 4 x 512 tokens in one pass 37 ms vs 49 ms
            # one by one, but 32 x 512 in one pass 335 ms vs 283 ms)
            group, tokens = [], 0
            for i in range(n + 1):
                if i == n or (group and tokens + len(prompts[i]) > self.PREFILL_TOKEN_BUDGET):
                    self.add_sequences_to([slots[j] for j in group], [prompts[j] for j in group],
                                          None if block_tables is None else [block_tables[j] for j in group])
                    group, tokens = [], 0
                if i < n:
                    group.append(i)
                    tokens += len(prompts[i])
            return
        for i, b in enumerate(slots):
            if block_tables is not None and block_tables[i] is not None:
                self.set_block_table(b, block_tables[i])
        lens = [len(p) for p in prompts]
        Ttot, Tmax = sum(lens), max(lens)
        s = Scratch(cfg, Ttot, n, Tmax, 1, dev, logits_rows=n)
        ids = torch.tensor([t for p in prompts for t in p], dtype=torch.int64, device=dev)
        pos = torch.cat([torch.arange(T, dtype=torch.int64, device=dev) for T in lens])
        slot_map = torch.cat([self._slots_for(b, torch.arange(T, dtype=torch.int64, device=dev)) for b, T in zip(slots, lens)])
        q_start = torch.tensor([0] + list(torch.tensor(lens).cumsum(0).tolist()), dtype=torch.int32, device=dev)
        md = AttentionMetadata(slot_map, self.block_tables[list(slots)].contiguous(),
                               torch.tensor(lens, dtype=torch.int32, device=dev), q_start, Tmax, 1)
        hs = self.model.forward(ids, pos, self.kv_caches, md, s, w4a4=False)
        last = (q_start[1:].long() - 1)
        logits = self.model.compute_logits(hs[last].contiguous(), s)
        probs = torch.empty(n, cfg.vocab_size, dtype=torch.float32, device=dev)
        tok = torch.empty(n, dtype=torch.int64, device=dev)
        self._first_token(logits, probs, tok, list(slots))
        for i, (b, T) in enumerate(zip(slots, lens)):
            self.seq_lens[b] = T + 1
            self.last_token[b] = tok[i]
            self.gen_tokens[b].fill_(-1)
            self.gen_tokens[b, 0] = tok[i]
            self.gen_lens[b] = 1
            self._len_ub[b] = T + 1
            self._gen_ub[b] = 1
        self.n_active = sum(1 for v in self._len_ub if v > 0)
        torch.cuda.synchronize()

    @torch.no_grad()
    def add_sequence(self, slot: int, prompt: Sequence[int], block_table: Optional[Sequence[int]] = None,
                     sync: bool = True) -> None:
        """Admit a request into an empty slot: scorer-only W4A16 prompt pass (the proposer never runs on prefill,
        spec_decode_worker.py:699); the first generated token is the target's greedy token."""
        cfg, dev, b = self.cfg, self.device, slot
        if not 0 <= b < self.B:
            raise ValueError(f"slot {b} outside the captured batch of {self.B}")
        if self._len_ub[b] != 0:
            raise ValueError(f"slot {b} is occupied; free_slot() it first")
        if block_table is not None:
            self.set_block_table(b, block_table)
        T = len(prompt)
        if T < 1 or T + 1 > min(self._capacity[b], self.max_model_len):
            raise ValueError(f"prompt of {T} tokens does not fit slot {b} (capacity {self._capacity[b]}, "
                             f"max_model_len {self.max_model_len})")
        if self._prefill_scratch is None or self._prefill_scratch.T < T:
            self._prefill_scratch = Scratch(cfg, T, 1, T, n_splits_for(T), dev, logits_rows=1)
        s = self._prefill_scratch
        ids = torch.tensor(prompt, dtype=torch.int64, device=dev)
        pos = torch.arange(T, dtype=torch.int64, device=dev)
        slots = self._slots_for(b, pos)
        md = AttentionMetadata(slots, self.block_tables[b:b + 1].contiguous(),
                               torch.tensor([T], dtype=torch.int32, device=dev),
                               torch.tensor([0, T], dtype=torch.int32, device=dev), T, n_splits_for(T))
        hs = self.model.forward(ids, pos, self.kv_caches, md, s, w4a4=False)
        logits = self.model.compute_logits(hs[T - 1:T], s)
        probs = torch.empty(1, cfg.vocab_size, dtype=torch.float32, device=dev)
        tok = torch.empty(1, dtype=torch.int64, device=dev)
        self._first_token(logits, probs, tok, [b])
        self.seq_lens[b] = T + 1
        self.last_token[b] = tok[0]
        self.gen_tokens[b].fill_(-1)
        self.gen_tokens[b, 0] = tok[0]
        self.gen_lens[b] = 1
        self._len_ub[b] = T + 1
        self._gen_ub[b] = 1
        self.n_active = sum(1 for v in self._len_ub if v > 0)
        if sync:
            torch.cuda.synchronize()

    def _first_token(self, logits, probs, tok, slots: List[int]) -> None:
        """The target's first token of freshly admitted prompts: greedy, or sampled with the slots' parameters."""
        if all(self._samp_host[b][0] < 1e-5 for b in slots):
            ops.softmax_argmax(logits, probs, tok)
            return
        idx = torch.tensor(slots, dtype=torch.int64, device=self.device)
        ops.sample_top_k_top_p(logits, probs, tok, self.samp_temp[idx].contiguous(), self.samp_topk[idx].contiguous(),
                               self.samp_topp[idx].contiguous(), rng_state=self.sampler.rng_state)

    def free_slot(self, slot: int) -> None:
        """The request in `slot` has finished (EOS / max_tokens / aborted): the slot goes back to empty."""
        self.seq_lens[slot] = 0
        self.gen_lens[slot] = 0
        self._len_ub[slot] = 0
        self._gen_ub[slot] = 0
        self.set_sampling_params(slot)          # back to greedy
        if self._bt_host[slot] is not None:
            # the request's blocks go back to the scheduler: the slot must not keep a table (and a capacity) that the
            # next admission could pass validation against.  Contiguous mode: the slot's own default range again;
            # brought-table mode (fewer blocks than slots x max_model_len): no table, capacity 0 until one arrives.
            if self.num_blocks >= self.B * self.blocks_per_seq:
                lo = slot * self.blocks_per_seq
                self.block_tables[slot].copy_(torch.arange(lo, lo + self.blocks_per_seq, dtype=torch.int32, device=self.device))
                self._capacity[slot] = self.blocks_per_seq * self.block_size
            else:
                self.block_tables[slot].zero_()
                self._capacity[slot] = 0
            self._bt_host[slot] = None
        self.n_active = sum(1 for v in self._len_ub if v > 0)

    def set_sampling_params(self, slot: int, temperature: float = 0.0, top_k: int = -1, top_p: float = 1.0) -> None:
        """The request in `slot` samples with these parameters (vllm/sampling_params.py; temperature < 1e-5 = greedy, for
        which vLLM itself resets top_k / top_p).  Uploaded only when they change."""
        if temperature < 1e-5:
            temperature, top_k, top_p = 0.0, -1, 1.0
        want = (float(temperature), int(top_k) if top_k and top_k > 0 else -1, float(top_p))
        if self._samp_host[slot] == want:
            return
        k1 = self.k + 1
        self.samp_temp[slot] = want[0]; self.samp_topk[slot] = want[1]; self.samp_topp[slot] = want[2]
        self.v_samp_temp[slot * k1:(slot + 1) * k1] = want[0]
        self.v_samp_topk[slot * k1:(slot + 1) * k1] = want[1]
        self.v_samp_topp[slot * k1:(slot + 1) * k1] = want[2]
        self._samp_host[slot] = want

    def _sampling_on(self) -> bool:
        return any(self._samp_host[b][0] >= 1e-5 for b in range(self.B) if self._len_ub[b] > 0)

    def set_block_table(self, slot: int, blocks: Sequence[int]) -> None:
        """The scheduler's block table of the request in `slot` (vLLM SequenceGroupMetadata.block_tables); it must
        cover every position a cycle can touch (vLLM allocates them as num_lookahead_slots).  The scheduler resends
        the table on every step: it is uploaded only when it differs from the one the slot already has."""
        blocks = [int(b) for b in blocks]
        if self._bt_host[slot] == blocks:
            return
        self._validate_block_table(blocks)
        n = len(blocks)
        row = torch.zeros(self.blocks_per_seq, dtype=torch.int32)
        row[:n] = torch.tensor(blocks, dtype=torch.int32)
        self.block_tables[slot].copy_(row.to(self.device))
        self._capacity[slot] = n * self.block_size
        self._bt_host[slot] = blocks

    def _validate_block_table(self, blocks: Sequence[int]) -> None:
        n = len(blocks)
        if n > self.blocks_per_seq:
            raise ValueError(f"{n} blocks > {self.blocks_per_seq} per sequence (max_model_len {self.max_model_len})")
        if n and (min(blocks) < 0 or max(blocks) >= self.num_blocks):
            raise ValueError("block id outside the KV cache")

    def active_slots(self) -> List[int]:
        return [b for b in range(self.B) if self._len_ub[b] > 0]

    def sync_lens(self) -> None:
        """Make the host-side length bounds exact (one small device read)."""
        both = torch.stack((self.seq_lens, self.gen_lens)).tolist()
        lens, gens = both[0], both[1]
        for b in range(self.B):
            if self._len_ub[b] > 0:
                self._len_ub[b], self._gen_ub[b] = int(lens[b]), int(gens[b])

    def note_emitted(self, emitted: Sequence[int]) -> None:
        """Exact bookkeeping from a cycle's output the caller has already read (the worker's one host read)."""
        for b in range(self.B):
            if self._len_before[b] > 0 and self._len_ub[b] > 0:
                self._len_ub[b] = self._len_before[b] + int(emitted[b])
                self._gen_ub[b] = self._gen_before[b] + int(emitted[b])

    def _check_capacity(self, per_step: int, slots: Optional[Sequence[int]] = None) -> None:
        cap_out = self.gen_tokens.shape[1]
        for b in (range(self.B) if slots is None else slots):
            L = self._len_ub[b]
            if L <= 0:
                continue
            # the cycle writes KV at positions L-1 .. L-1+k and then knows up to L+k+1 tokens
            if L - 1 + per_step > min(self._capacity[b], self.max_model_len):
                raise ValueError(f"slot {b}: a cycle from length {L} (upper bound) needs positions up to {L - 1 + per_step - 1} "
                                 f"but the sequence holds {min(self._capacity[b], self.max_model_len)} "
                                 "(block table / max_model_len): finish the request or extend its block table")
            if self._gen_ub[b] + per_step > cap_out:
                raise ValueError(f"slot {b}: output buffer of {cap_out} tokens is full (max_new_tokens)")

    def _slots_for(self, b: int, pos: torch.Tensor) -> torch.Tensor:
        bt = self.block_tables[b].to(torch.int64)
        return (bt[pos // self.block_size] * self.block_size + pos % self.block_size).contiguous()

    # ------------------------------------------------------------------ one speculative cycle (:758-858)
    def _snapshot(self):
        # the small sequence state aside: what recover() restarts the cycle from if a device-side hand-off timed out
        ops.spec_snapshot(self.seq_lens, self.gen_lens, self.last_token, self.sampler.counters, self.sampler.rng_state,
                          self._snap_i32, self._snap_i64)

    def _cycle_body(self):
        self._snapshot()
        self._draft_body()
        self._verify_body()
        self._collect_errors()

    def _comm(self):
        tp = getattr(self.model, "tp", None)
        return tp if tp is not None and tp.world > 1 else None

    def _collect_errors(self):
        """Last launches of a cycle: err_word = OR of the sticky error words of this stream's hand-off workspaces and of
        the one-shot all-reduce, summed over the ranks under tensor parallelism -- every rank reads the SAME word with
        its output tokens, so the decision to re-run (or to give up) is collective by construction."""
        tp = self._comm()
        extra = []
        if tp is not None and hasattr(tp.comm, "error_word_address"):
            extra.append(tp.comm.error_word_address())
        ops.collect_error_words(ops.stream_error_words(self.device, extra), out=self.err_word)
        if tp is not None:
            tp.all_reduce(self.err_word)

    def _draft_body(self):
        m, k, B, bs = self.model, self.k, self.B, self.block_size
        # proposer: k draft steps, W4A4  (execute_model_req.w4a4 = True, :799)
        # (the lengths of the slots taking part, the first draft step's inputs and its embedding rows: one launch)
        emb_d = (m.embed_tokens, self.scratch_draft.hidden[:B]) if self.EMBED_IN_BOOKKEEPING else None
        if emb_d is not None:
            ops.spec_prepare_draft(self.last_token, self.seq_lens, self.block_tables, bs, self.d_tokens, self.d_pos,
                                   self.d_slots, self.d_ctx, embed=emb_d, step_mask=self.step_mask, eff_lens=self.eff_lens)
        else:
            torch.mul(self.seq_lens, self.step_mask, out=self.eff_lens)
            ops.spec_prepare_draft(self.last_token, self.eff_lens, self.block_tables, bs, self.d_tokens, self.d_pos,
                                   self.d_slots, self.d_ctx)
        tp = self._comm()
        draft_sv = tp is not None and getattr(tp, "shard_draft_vocab", False)   # vocab-parallel lm_head on the draft pass too
        for i in range(k):
            hs = m.forward(self.d_tokens, self.d_pos, self.kv_caches, self.md_draft, self.scratch_draft, w4a4=True,
                           embedded=emb_d is not None)
            if self._mode_sampling:   # draft tokens are SAMPLED from the processed draft distribution (sampler.py:216-316)
                m.sample(hs, self.scratch_draft, self.draft_probs_kbv[i], self.draft_ids_kb[i], self.samp_temp, self.samp_topk,
                         self.samp_topp, self.sampler.rng_state, shard_vocab=draft_sv)
            else:
                m.sample_greedy(hs, self.scratch_draft, self.draft_probs_kbv[i], self.draft_ids_kb[i], shard_vocab=draft_sv)
            if i != k - 1:  # _gpu_advance_step (draft_model_runner.py:78-135)
                ops.spec_advance_draft(bs, self.d_tokens, self.draft_ids_kb[i], self.d_pos, self.d_ctx, self.d_slots,
                                       self.block_tables, embed=emb_d)

    def _verify_body(self):
        m, k, B, bs = self.model, self.k, self.B, self.block_size
        # scorer: one W4A16 pass over [last, d_1..d_k] per sequence  (w4a4 = False, :812; mqa_scorer.py)
        draft_ids = self.draft_ids_kb.transpose(0, 1)            # [B,k] view
        draft_probs = self.draft_probs_kbv.transpose(0, 1)       # [B,k,V] view
        emb_v = (m.embed_tokens, self.scratch_verify.hidden[:B * (k + 1)]) if self.EMBED_IN_BOOKKEEPING else None
        ops.spec_prepare_verify(self.last_token, draft_ids, self.eff_lens, self.block_tables, bs, self.v_tokens,
                                self.v_pos, self.v_slots, self.v_ctx, embed=emb_v)
        hs = m.forward(self.v_tokens, self.v_pos, self.kv_caches, self.md_verify, self.scratch_verify, w4a4=False,
                       embedded=emb_v is not None)
        hook = self._verify_logits_hook(draft_ids)
        if self._mode_sampling:
            m.sample(hs, self.scratch_verify, self.target_probs.view(B * (k + 1), -1), self.target_tokens.view(-1),
                     self.v_samp_temp, self.v_samp_topk, self.v_samp_topp, self.sampler.rng_state, shard_vocab=True,
                     logits_hook=hook)
        else:
            m.sample_greedy(hs, self.scratch_verify, self.target_probs.view(B * (k + 1), -1), self.target_tokens.view(-1),
                            shard_vocab=True, logits_hook=hook)
        # _verify_tokens (:861-970): bonus = the target's own token at the last position
        self.sampler.forward(self.target_probs, self.target_tokens[:, k], draft_probs, draft_ids, out=self.out_tokens,
                             accepted=self.accepted, recovered=self.recovered, uniform=self.inject_uniform,
                             exponential=self.inject_exponential, active_lens=self.eff_lens)
        # _create_output_sampler_list bookkeeping (:972-1063)
        ops.spec_commit(self.out_tokens, self.seq_lens, self.last_token, self.gen_tokens, self.gen_lens)

    @torch.no_grad()
    def _verify_logits_hook(self, draft_ids):
        """A callable(logits) applied to the verify logits inside the cycle, or None.  The product has none; bench.py's
        BenchEngine overrides this (synthetic draft/target agreement).  A method, not a stored closure: an engine must
        not sit in a reference cycle -- its hipGraphs would then be destroyed by the cyclic collector at an arbitrary
        moment, and destroying a graph while another capture is under way aborts the process."""
        return None

    def _set_participants(self, participants: Optional[Sequence[int]]) -> List[int]:
        want = [1] * self.B if participants is None else [1 if b in participants else 0 for b in range(self.B)]
        if want != self._mask_host:
            self.step_mask.copy_(torch.tensor(want, dtype=torch.int32), non_blocking=False)
            self._mask_host = want
        return [b for b in range(self.B) if want[b] and self._len_ub[b] > 0]

    def step_no_spec(self, participants: Optional[Sequence[int]] = None):
        """One NON-speculative decode step (spec_decode_worker.py:666-720 on a decode batch: the scorer alone, W4A16,
        one token per sequence): taken when speculation is disabled for a step (`num_lookahead_slots == 0`,
        `speculative_disable_by_batch_size`).  Eager: it is the rare path, the captured graph is the cycle's."""
        m, bs = self.model, self.block_size
        on = self._set_participants(participants)
        self._check_capacity(1, on)
        self._len_before, self._gen_before = list(self._len_ub), list(self._gen_ub)
        torch.mul(self.seq_lens, self.step_mask, out=self.eff_lens)
        ops.spec_prepare_draft(self.last_token, self.eff_lens, self.block_tables, bs, self.d_tokens, self.d_pos,
                               self.d_slots, self.d_ctx)                   # [last token] at position seq_len - 1
        hs = m.forward(self.d_tokens, self.d_pos, self.kv_caches, self.md_draft, self.scratch_draft, w4a4=False)
        logits = m.compute_logits(hs, self.scratch_draft)
        ops.softmax_argmax(logits, self.draft_probs_kbv[0], self.draft_ids_kb[0])
        self.out_tokens.fill_(-1)
        self.out_tokens[:, 0] = torch.where(self.eff_lens > 0, self.draft_ids_kb[0], torch.full_like(self.draft_ids_kb[0], -1))
        ops.spec_commit(self.out_tokens, self.seq_lens, self.last_token, self.gen_tokens, self.gen_lens)
        for b in on:
            self._len_ub[b] += 1
            self._gen_ub[b] += 1

    @torch.no_grad()
    def step(self, participants: Optional[Sequence[int]] = None):
        """Enqueue one cycle on the current stream (graph replay after the first call).  Raises ValueError -- before
        anything is enqueued -- if a sequence could outgrow its block table, max_model_len or the output buffer.
        participants: the slots taking part (default: every occupied slot); the others keep their state."""
        on = self._set_participants(participants)
        self._check_capacity(self.k + 1, on)
        self._len_before, self._gen_before = list(self._len_ub), list(self._gen_ub)
        for b in on:                       # until the caller reports / syncs the real numbers: the worst case
            self._len_ub[b] += self.k + 1
            self._gen_ub[b] += self.k + 1
        self._mode_sampling = self._sampling_on()     # (every rank sees the same requests: the same decision)
        if not self.use_graph:
            self._cycle_body()
            return
        full, draft = (self._graph_s, self._graph_draft_s) if self._mode_sampling else (self._graph, self._graph_draft)
        if full is None and draft is None:
            self._capture()
            if not self.use_graph:
                self._cycle_body()
                return
            full, draft = (self._graph_s, self._graph_draft_s) if self._mode_sampling else (self._graph, self._graph_draft)
        if full is not None:
            full.replay()
        else:   # the verify pass could not be captured (its collectives): proposer from its graph, scorer eagerly.
            # The same bracket as _cycle_body: the state snapshot sits at the head of the draft graph (_capture), the
            # error words are collected behind the eager verify pass -- read_outputs() / recover() work in this mode too
            draft.replay()
            self._verify_body()
            self._collect_errors()

    def _capture(self):
        # warm up outside capture (lazy module loads, LDS attribute), on a side stream as torch requires.
        # The warm-up cycles advance the sequence state; restore it so capture does not consume tokens.
        state = [t.clone() for t in (self.seq_lens, self.last_token, self.gen_tokens, self.gen_lens,
                                     self.sampler.counters, self.sampler.rng_state)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._cycle_body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        import gc
        gc.collect()   # no stale CUDAGraph may be finalised while this capture is under way
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._cycle_body()
            torch.cuda.synchronize()
        except Exception as exc:  # e.g. a collective backend that cannot be captured: loudly, not silently
            import warnings
            torch.cuda.synchronize()
            for t, s in zip((self.seq_lens, self.last_token, self.gen_tokens, self.gen_lens, self.sampler.counters,
                             self.sampler.rng_state), state):
                t.copy_(s)
            try:   # the proposer has no collectives (replicated under TP): keep its k forwards in a graph
                gd = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gd):
                    self._snapshot()
                    self._draft_body()
                torch.cuda.synchronize()
                if self._mode_sampling:
                    self._graph_draft_s = gd
                else:
                    self._graph_draft = gd
                warnings.warn(f"hipGraph capture of the whole cycle failed ({exc!r}); "
                              "the draft pass replays from a graph, the verify pass runs eagerly")
            except Exception as exc2:
                warnings.warn(f"hipGraph capture failed ({exc!r}; draft only: {exc2!r}); running the cycle eagerly")
                self.use_graph = False
                torch.cuda.synchronize()
            for t, s in zip((self.seq_lens, self.last_token, self.gen_tokens, self.gen_lens, self.sampler.counters,
                             self.sampler.rng_state), state):
                t.copy_(s)
            return
        for t, s in zip((self.seq_lens, self.last_token, self.gen_tokens, self.gen_lens, self.sampler.counters,
                         self.sampler.rng_state), state):
            t.copy_(s)
        if self._mode_sampling:
            self._graph_s = g
        else:
            self._graph = g

    # ------------------------------------------------------------------ results
    def generated(self) -> List[List[int]]:
        torch.cuda.synchronize()
        lens = self.gen_lens.tolist()
        toks = self.gen_tokens.cpu()
        return [toks[b, :lens[b]].tolist() for b in range(self.B)]

    def metrics(self) -> SpecDecodeWorkerMetrics:
        a, e, d = (int(v) for v in self.sampler.counters.tolist())
        return metrics_from_counters(a, e, d, self.k)

    def max_cycles_left(self) -> int:
        """Cycles that can still run before step() would refuse (exact lengths: syncs)."""
        self.sync_lens()
        room = []
        for b in self.active_slots():
            cap = min(self._capacity[b], self.max_model_len)
            room.append((cap - self._len_ub[b] + 1) // (self.k + 1))
            room.append((self.gen_tokens.shape[1] - self._gen_ub[b]) // (self.k + 1))
        return max(0, min(room)) if room else 0

    def read_outputs(self):
        """The worker's ONE host read per cycle: (out_tokens [B, k+1] on the host, error word).  A non-zero error word
        means a device-side hand-off (spread Hadamard, norm hand-off, one-shot all-reduce) timed out somewhere in the
        cycle on some rank: the tokens are invalid; see recover()."""
        self._out_err_host.copy_(self._out_err, non_blocking=True)
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        host = self._out_err_host
        return host[:-1].view(self.B, self.k + 1), int(host[-1])

    @torch.no_grad()
    def recover(self) -> None:
        """Re-run the cycle that has just failed (read_outputs() returned a non-zero error word) from the state
        snapshot taken at its start, eagerly and WITHOUT the kernels that wait for partner workgroups: the spread MLP
        Hadamard and the norm hand-off fall back to their one-workgroup / recompute forms, which are bit-identical
        (tests/test_kernels_gpu.py), so the replayed cycle emits exactly what the failed one should have.  The KV slots
        the failed cycle wrote are the ones the replay rewrites.  The caller reads the outputs again; a second failure
        (e.g. a lost tensor-parallel peer) is final."""
        ops.spec_snapshot(self.seq_lens, self.gen_lens, self.last_token, self.sampler.counters, self.sampler.rng_state,
                          self._snap_i32, self._snap_i64, restore=True)
        self.clear_error_words()
        saved = ops.XWG_SPREAD, ops.LN_HANDOFF
        ops.XWG_SPREAD = ops.LN_HANDOFF = False
        try:
            self._cycle_body()
        finally:
            ops.XWG_SPREAD, ops.LN_HANDOFF = saved
        self.recoveries += 1

    def clear_error_words(self) -> None:
        dev = self.device
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        for key, ws in ops._xwg_ws.items():
            if key[:2] == (dev.type, idx):
                ws[:1].zero_()
        for key, ws in ops._ln_ws.items():
            if key[:2] == (dev.type, idx):
                ws.view(torch.int32)[31:32].zero_()
        tp = self._comm()
        if tp is not None and hasattr(tp.comm, "error_word_address"):
            ops.collect_error_words([tp.comm.error_word_address()], clear=True)
        self.err_word.zero_()

    def error_flag(self) -> int:
        """Sticky device-side error words of EVERY hand-off workspace of the device and of the one-shot all-reduce, read
        from the host (0 = fine).  The cycle itself reports through read_outputs(); this is the out-of-band check."""
        words = [w for w in (ops.xwg_error_word(self.device), ops.ln_linear_error_word(self.device)) if w is not None]
        flag = int(torch.cat(words).abs().max().item()) if words else 0
        tp = self._comm()
        if tp is not None and hasattr(tp.comm, "error"):
            flag |= int(tp.comm.error())
        return flag
