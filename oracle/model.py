"""CPU restatement of the whole QSpec model forward and of one speculative cycle, composed from the
primitives of oracle/__init__.py (TEST INFRASTRUCTURE ONLY -- see that module's header).

Follows vllm/model_executor/models/quarot_llama.py op for op:
  QuarotDecoderLayer.forward :363-392, QuarotLlamaAttention.forward :177-243, QuarotLlamaMLP.forward :266-299,
  LlamaModel.forward :484-535 (final norm always fp16, :533), lm_head (logits_processor.py:92-97),
and vllm/spec_decode/spec_decode_worker.py:758-1063 for the cycle.
Weights are numpy arrays in the checkpoint layout (packed int8 [N,K/2], fp16 scales).
"""
from __future__ import annotations

import numpy as np

import oracle as O


class OracleModel:
    def __init__(self, cfg, layers, embed_tokens, lm_head, had_rem_dim, had_K, cos_sin_cache, block_size,
                 head_had=None, head_had_K=1):
        """layers: list of dicts {qkv_w,qkv_s,o_w,o_s,gate_up_w,gate_up_s,down_w,down_s} (numpy)."""
        self.cfg, self.layers = cfg, layers
        self.embed_tokens, self.lm_head = embed_tokens, lm_head
        self.had, self.had_K = had_rem_dim, had_K
        self.head_had, self.head_had_K = head_had, head_had_K      # table factor of get_hadK(num_heads), if any
        self.cs = cos_sin_cache
        self.block_size = block_size
        self.head_scale = O.rsqrt_scale(cfg.num_attention_heads)
        self.mlp_scale = O.rsqrt_scale(cfg.intermediate_size)
        self.sm_scale = cfg.head_dim ** -0.5
        self.w4a16_f32acc = False    # noise-floor probe: W4A16 GEMMs through the second admissible implementation

    @classmethod
    def from_torch_model(cls, m, block_size, max_layers=None):
        """Copy the weights of a qspec_amd.model.QuarotLlamaForCausalLM to the host (same bytes)."""
        c = lambda t: t.detach().cpu().numpy()  # noqa: E731
        layers = [dict(qkv_w=c(l.qkv_proj.weight), qkv_s=c(l.qkv_proj.weight_scales).reshape(-1),
                       o_w=c(l.o_proj.weight), o_s=c(l.o_proj.weight_scales).reshape(-1),
                       gate_up_w=c(l.gate_up.weight), gate_up_s=c(l.gate_up.weight_scales).reshape(-1),
                       down_w=c(l.down_proj.weight), down_s=c(l.down_proj.weight_scales).reshape(-1))
                  for l in (m.layers if max_layers is None else m.layers[:max_layers])]
        had = c(m.had_rem_dim) if m.had_rem_dim is not None else None
        hh = c(m.head_had) if getattr(m, "head_had", None) is not None else None
        return cls(m.config, layers, c(m.embed_tokens), c(m.lm_head), had, m.had_K, c(m.cos_sin_cache), block_size,
                   hh, getattr(m, "head_had_K", 1))

    def _linear(self, x, w, s, w4a4):
        if w4a4:
            q, sc = x
            return O.gemm_w4a4(q, sc, w, s)
        return O.gemm_w4a16_f32acc(x, w, s) if self.w4a16_f32acc else O.gemm_w4a16(x, w, s)

    def forward(self, input_ids, positions, kv_caches, slot_mapping, block_tables, ctx_lens, q_start, w4a4,
                return_trace=False):
        """kv_caches: list of (key_cache, value_cache) numpy [num_blocks, block_size, n_kv, d], updated in place."""
        cfg = self.cfg
        eps = cfg.rms_norm_eps
        T = len(input_ids)
        nq, nkv, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        hidden = self.embed_tokens[np.asarray(input_ids)].copy()
        trace = {}
        for li, L in enumerate(self.layers):
            kc, vc = kv_caches[li]
            if w4a4:
                q, s, _ = O.ln_quant_i4(hidden, eps)
                x = (q, s)
            else:
                x = O.ln_fp16(hidden, eps)
            qkv = self._linear(x, L["qkv_w"], L["qkv_s"], w4a4)
            qr, kr = O.rope_neox(positions, qkv[:, :nq * d], qkv[:, nq * d:(nq + nkv) * d], self.cs, d)
            v = qkv[:, (nq + nkv) * d:]
            O.reshape_and_cache_flash(kr.reshape(T, nkv, d), v.reshape(T, nkv, d), kc, vc, slot_mapping)
            attn = O.paged_attention(qr, kc, vc, block_tables, ctx_lens, q_start, self.sm_scale)
            a = O.heads_hadamard(attn, nq, self.head_scale, self.head_had, self.head_had_K)
            if w4a4:
                a = O.rowabsmax_quant_i4(a, 1.0)
            o = self._linear(a, L["o_w"], L["o_s"], w4a4)
            if return_trace:   # per-layer intermediates for teacher-forced tests (input = the previous hidden_{li-1})
                trace[f"qkv_{li}"] = np.concatenate([qr, kr, v], axis=1)
                trace[f"attn_{li}"] = attn
                trace[f"o_in_{li}"] = a          # (q, scale) under w4a4, fp16 otherwise
                trace[f"o_out_{li}"] = o
                trace[f"ln1_{li}"] = x           # the layer's first norm: (q, scale) under w4a4, fp16 otherwise
            hidden = O.add_f16(hidden, o)
            if return_trace:
                trace[f"hidden_attn_{li}"] = hidden.copy()
            if w4a4:
                q, s, _ = O.ln_quant_i4(hidden, eps)
                x = (q, s)
            else:
                x = O.ln_fp16(hidden, eps)
            gu = self._linear(x, L["gate_up_w"], L["gate_up_s"], w4a4)
            act = O.silu_mul(gu, cfg.intermediate_size)
            g = O.mlp_hadamard(act, self.had, self.had_K, self.mlp_scale)
            if w4a4:
                g = O.rowabsmax_quant_i4(g, 1.0)
            if return_trace:
                trace[f"ln2_{li}"] = x
                trace[f"act_{li}"] = act
                trace[f"down_in_{li}"] = g
            dn = self._linear(g, L["down_w"], L["down_s"], w4a4)
            if return_trace:
                trace[f"down_out_{li}"] = dn
            hidden = O.add_f16(hidden, dn)
            if return_trace:
                trace[f"hidden_{li}"] = hidden.copy()
        out = O.ln_fp16(hidden, eps)
        return (out, trace) if return_trace else out

    def logits(self, hidden):
        return O.gemm_f16(hidden, self.lm_head)


class OracleEngine:
    """The cycle of qspec_amd.spec_decode.engine.QSpecEngine on the CPU (same state layout)."""

    def __init__(self, model: OracleModel, k, B, max_model_len, block_size):
        cfg = model.cfg
        self.m, self.k, self.B, self.bs = model, k, B, block_size
        self.blocks_per_seq = (max_model_len + block_size - 1) // block_size
        nb = B * self.blocks_per_seq
        shape = (nb, block_size, cfg.num_key_value_heads, cfg.head_dim)
        self.kv = [(np.zeros(shape, np.float16), np.zeros(shape, np.float16)) for _ in range(cfg.num_hidden_layers)]
        self.block_tables = np.arange(nb, dtype=np.int32).reshape(B, self.blocks_per_seq)
        self.seq_lens = np.zeros(B, np.int32)
        self.last_token = np.zeros(B, np.int64)
        self.generated = [[] for _ in range(B)]
        self.counters = [0, 0, 0]
        self.prefill_probs = [None] * B   # the target's distribution at the end of each prompt (near-tie diagnostics)
        self.agreement_rho = None  # bench-only synthetic agreement (see qspec_bench_force_agreement)
        self._agree_rng = np.random.default_rng(1234)

    def _slots(self, b, pos):
        pos = np.asarray(pos)
        return self.block_tables[b, pos // self.bs].astype(np.int64) * self.bs + pos % self.bs

    def add_sequences(self, prompts):
        for b, p in enumerate(prompts):
            T = len(p)
            pos = np.arange(T, dtype=np.int64)
            hs = self.m.forward(np.asarray(p), pos, self.kv, self._slots(b, pos), self.block_tables[b:b + 1],
                                np.array([T], np.int32), np.array([0, T], np.int32), w4a4=False)
            pr, tok = O.softmax_argmax(self.m.logits(hs[T - 1:T]))
            self.prefill_probs[b] = pr[0]
            self.seq_lens[b] = T + 1
            self.last_token[b] = tok[0]
            self.generated[b].append(int(tok[0]))

    def step(self, uniform, exponential, forced_draft_ids=None, forced_out=None):
        """uniform [B,k], exponential [B,k,V]: the random draws of the rejection sampler (injected).
        forced_draft_ids [B,k]: teacher forcing -- feed these tokens instead of the oracle's own argmax (its free
        choice is still returned as draft_ids_free); forced_out [B,k+1]: commit this output instead of the
        oracle's, so that the state follows an implementation under test."""
        m, k, B = self.m, self.k, self.B
        V = m.cfg.vocab_size
        tokens = self.last_token.copy()
        pos = (self.seq_lens - 1).astype(np.int64)
        ctx = self.seq_lens.copy()
        qs1 = np.arange(B + 1, dtype=np.int32)
        draft_probs = np.zeros((B, k, V), np.float32)
        draft_ids = np.zeros((B, k), np.int64)
        draft_ids_free = np.zeros((B, k), np.int64)
        for i in range(k):
            slots = np.array([self._slots(b, pos[b]) for b in range(B)], np.int64)
            hs = m.forward(tokens, pos, self.kv, slots, self.block_tables, ctx, qs1, w4a4=True)
            p, t = O.softmax_argmax(m.logits(hs))
            draft_ids_free[:, i] = t
            if forced_draft_ids is not None:
                t = np.asarray(forced_draft_ids)[:, i].astype(np.int64)
            draft_probs[:, i], draft_ids[:, i] = p, t
            tokens, pos, ctx = t.copy(), pos + 1, ctx + 1
        vt = np.concatenate([np.concatenate([[self.last_token[b]], draft_ids[b]]) for b in range(B)])
        vp = np.concatenate([np.arange(self.seq_lens[b] - 1, self.seq_lens[b] + k) for b in range(B)]).astype(np.int64)
        vs = np.concatenate([self._slots(b, np.arange(self.seq_lens[b] - 1, self.seq_lens[b] + k)) for b in range(B)])
        hs = m.forward(vt, vp, self.kv, vs, self.block_tables, (self.seq_lens + k).astype(np.int32),
                       (np.arange(B + 1) * (k + 1)).astype(np.int32), w4a4=False)
        logits = m.logits(hs)
        if self.agreement_rho is not None:
            for b in range(B):
                for i in range(k):
                    if self._agree_rng.random() < self.agreement_rho:
                        logits[b * (k + 1) + i, draft_ids[b, i]] = np.float16(60000.0)
        tp, tt = O.softmax_argmax(logits)
        tp, tt = tp.reshape(B, k + 1, V), tt.reshape(B, k + 1)
        out, accepted, recovered, c = O.rejection_sample(tp, tt[:, k], draft_probs, draft_ids, uniform, exponential)
        if forced_out is not None:
            out = np.asarray(forced_out)
        for j in range(3):
            self.counters[j] += c[j]
        for b in range(B):
            em = [int(x) for x in out[b] if x != -1]
            self.generated[b].extend(em)
            self.seq_lens[b] += len(em)
            self.last_token[b] = em[-1]
        return dict(out=out, accepted=accepted, recovered=recovered, draft_ids=draft_ids,
                    draft_ids_free=draft_ids_free, draft_probs=draft_probs,
                    target_probs=tp, target_tokens=tt)
