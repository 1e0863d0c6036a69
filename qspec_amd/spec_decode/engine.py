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
        return max(1, min(16, (256 + n_groups - 1) // n_groups))
    return 1


class QSpecEngine:
    def __init__(self, model: QuarotLlamaForCausalLM, num_speculative_tokens: int = 3, max_batch: int = 4,
                 max_model_len: int = 1024, block_size: int = 16, max_new_tokens: int = 1024, use_graph: bool = True,
                 seed: int = 0):
        self.model = model
        self.cfg = cfg = model.config
        self.k = k = num_speculative_tokens
        self.B = B = max_batch
        self.block_size = block_size
        self.max_model_len = max_model_len
        self.use_graph = use_graph
        dev = self.device = model.device
        i64, i32 = torch.int64, torch.int32
        # ---- one KV cache for both passes: [num_blocks, block_size, n_kv, d] per layer
        self.blocks_per_seq = (max_model_len + block_size - 1) // block_size
        self.num_blocks = B * self.blocks_per_seq
        shape = (self.num_blocks, block_size, cfg.num_key_value_heads, cfg.head_dim)
        self.kv_caches = [(torch.zeros(shape, dtype=torch.float16, device=dev),
                           torch.zeros(shape, dtype=torch.float16, device=dev)) for _ in range(cfg.num_hidden_layers)]
        self.block_tables = torch.arange(self.num_blocks, dtype=i32, device=dev).view(B, self.blocks_per_seq).contiguous()
        # ---- sequence state
        self.seq_lens = torch.zeros(B, dtype=i32, device=dev)      # L: tokens known (KV valid below L-1)
        self.last_token = torch.zeros(B, dtype=i64, device=dev)
        self.gen_tokens = torch.full((B, max_new_tokens + k + 1), -1, dtype=i64, device=dev)
        self.gen_lens = torch.zeros(B, dtype=i32, device=dev)
        self.n_active = 0
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
        self.out_tokens = torch.full((B, k + 1), -1, dtype=i64, device=dev)
        self.accepted = torch.zeros(B, k, dtype=torch.uint8, device=dev)
        self.recovered = torch.zeros(B, k, dtype=i64, device=dev)
        self.sampler = RejectionSampler(seed=seed)
        self.sampler.init_gpu_tensors(str(dev))
        self.scratch_draft = Scratch(cfg, B, B, 1, n_splits, dev)
        self.scratch_verify = Scratch(cfg, T, B, k + 1, n_splits, dev)
        self.md_draft = AttentionMetadata(self.d_slots, self.block_tables, self.d_ctx, self.d_qstart, 1, n_splits)
        self.md_verify = AttentionMetadata(self.v_slots, self.block_tables, self.v_ctx, self.v_qstart, k + 1, n_splits)
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._graph_draft: Optional[torch.cuda.CUDAGraph] = None   # fallback: proposer only (verify pass eager)
        # test hooks: injected random draws for the rejection sampler (eager mode only)
        self.inject_uniform: Optional[torch.Tensor] = None
        self.inject_exponential: Optional[torch.Tensor] = None
        self._prefill_scratch: Optional[Scratch] = None
        # bench-only synthetic agreement between draft and target (None = the weights' own agreement)
        self.agreement_rho: Optional[float] = None

    # ------------------------------------------------------------------ prefill (_run_no_spec, :666-720)
    @torch.no_grad()
    def add_sequences(self, prompts: Sequence[Sequence[int]]):
        """Prompt pass: scorer only, W4A16 (the proposer never runs on prefill, spec_decode_worker.py:699);
        the first generated token is the target's greedy token."""
        assert len(prompts) == self.B, "the cycle graph is captured for a fixed batch"
        cfg, dev = self.cfg, self.device
        for b, prompt in enumerate(prompts):
            T = len(prompt)
            assert 0 < T + self.k + 2 <= self.max_model_len
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
            ops.softmax_argmax(logits, probs, tok)
            self.seq_lens[b] = T + 1
            self.last_token[b] = tok[0]
            self.gen_tokens[b, 0] = tok[0]
            self.gen_lens[b] = 1
        self.n_active = self.B
        torch.cuda.synchronize()

    def _slots_for(self, b: int, pos: torch.Tensor) -> torch.Tensor:
        bt = self.block_tables[b].to(torch.int64)
        return (bt[pos // self.block_size] * self.block_size + pos % self.block_size).contiguous()

    # ------------------------------------------------------------------ one speculative cycle (:758-858)
    def _cycle_body(self):
        self._draft_body()
        self._verify_body()

    def _draft_body(self):
        m, k, B, bs = self.model, self.k, self.B, self.block_size
        # proposer: k draft steps, W4A4  (execute_model_req.w4a4 = True, :799)
        ops.spec_prepare_draft(self.last_token, self.seq_lens, self.block_tables, bs, self.d_tokens, self.d_pos,
                               self.d_slots, self.d_ctx)
        for i in range(k):
            hs = m.forward(self.d_tokens, self.d_pos, self.kv_caches, self.md_draft, self.scratch_draft, w4a4=True)
            logits = m.compute_logits(hs, self.scratch_draft)
            ops.softmax_argmax(logits, self.draft_probs_kbv[i], self.draft_ids_kb[i])
            if i != k - 1:  # _gpu_advance_step (draft_model_runner.py:78-135)
                ops.advance_step_flashattn(B, B, bs, self.d_tokens, self.draft_ids_kb[i], self.d_pos, self.d_ctx,
                                           self.d_slots, self.block_tables)

    def _verify_body(self):
        m, k, B, bs = self.model, self.k, self.B, self.block_size
        # scorer: one W4A16 pass over [last, d_1..d_k] per sequence  (w4a4 = False, :812; mqa_scorer.py)
        draft_ids = self.draft_ids_kb.transpose(0, 1)            # [B,k] view
        draft_probs = self.draft_probs_kbv.transpose(0, 1)       # [B,k,V] view
        ops.spec_prepare_verify(self.last_token, draft_ids, self.seq_lens, self.block_tables, bs, self.v_tokens,
                                self.v_pos, self.v_slots, self.v_ctx)
        hs = m.forward(self.v_tokens, self.v_pos, self.kv_caches, self.md_verify, self.scratch_verify, w4a4=False)
        logits = m.compute_logits(hs, self.scratch_verify, shard_vocab=True)
        if self.agreement_rho is not None:
            ops.bench_force_agreement(logits, draft_ids, self.agreement_rho, self.sampler.rng_state)
        ops.softmax_argmax(logits, self.target_probs.view(B * (k + 1), -1), self.target_tokens.view(-1))
        # _verify_tokens (:861-970): bonus = the target's own token at the last position
        self.sampler.forward(self.target_probs, self.target_tokens[:, k], draft_probs, draft_ids, out=self.out_tokens,
                             accepted=self.accepted, recovered=self.recovered, uniform=self.inject_uniform,
                             exponential=self.inject_exponential)
        # _create_output_sampler_list bookkeeping (:972-1063)
        ops.spec_commit(self.out_tokens, self.seq_lens, self.last_token, self.gen_tokens, self.gen_lens)

    @torch.no_grad()
    def step_no_spec(self):
        """One NON-speculative decode step (spec_decode_worker.py:666-720 on a decode batch: the scorer alone, W4A16,
        one token per sequence): taken when speculation is disabled for a step (`num_lookahead_slots == 0`,
        `speculative_disable_by_batch_size`).  Eager: it is the rare path, the captured graph is the cycle's."""
        m, bs = self.model, self.block_size
        ops.spec_prepare_draft(self.last_token, self.seq_lens, self.block_tables, bs, self.d_tokens, self.d_pos,
                               self.d_slots, self.d_ctx)                   # [last token] at position seq_len - 1
        hs = m.forward(self.d_tokens, self.d_pos, self.kv_caches, self.md_draft, self.scratch_draft, w4a4=False)
        logits = m.compute_logits(hs, self.scratch_draft)
        ops.softmax_argmax(logits, self.draft_probs_kbv[0], self.draft_ids_kb[0])
        self.out_tokens.fill_(-1)
        self.out_tokens[:, 0] = self.draft_ids_kb[0]
        ops.spec_commit(self.out_tokens, self.seq_lens, self.last_token, self.gen_tokens, self.gen_lens)

    @torch.no_grad()
    def step(self):
        """Enqueue one cycle on the current stream (graph replay after the first call)."""
        if not self.use_graph:
            self._cycle_body()
            return
        if self._graph is None and self._graph_draft is None:
            self._capture()
            if not self.use_graph:
                self._cycle_body()
                return
        if self._graph is not None:
            self._graph.replay()
        else:   # the verify pass could not be captured (its collectives): proposer from its graph, scorer eagerly
            self._graph_draft.replay()
            self._verify_body()

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
                    self._draft_body()
                torch.cuda.synchronize()
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
        """Cycles that can still run before the longest sequence could overflow its blocks / output buffer."""
        L = int(self.seq_lens.max().item())
        room_ctx = (self.max_model_len - L - 1) // (self.k + 1)
        room_out = (self.gen_tokens.shape[1] - int(self.gen_lens.max().item())) // (self.k + 1) - 1
        return max(0, min(room_ctx, room_out))
