"""QuarotLlamaForCausalLM on MI355X: the model shared by the draft (W4A4) and verify (W4A16) passes.

Mirror of vllm/model_executor/models/quarot_llama.py (QuarotLlamaAttention :62, QuarotLlamaMLP :247,
QuarotDecoderLayer :319, LlamaModel :436, QuarotLlamaForCausalLM :597): same weights, same op order, same
`forward(input_ids, positions, kv_caches, attn_metadata, **kwargs)` with the `w4a4` kwarg selecting the pass.
One set of packed int4 weights and one paged KV cache serve both passes.

Two implementations of the layer body:
  * `forward`            -- the product path: 7 (draft) / 9 (verify) fused HIP launches per layer, no transposes / copies;
  * `forward_modulewise` -- module by module through qspec_amd.quarot_nn exactly in the reference's order
                            (one reference op per call); kept for parity tests of the fusion.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import hadamard_tables, ops, quarot_nn


@dataclass
class QuarotLlamaConfig:
    hidden_size: int = 4096
    intermediate_size: int = 14336
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    num_hidden_layers: int = 32
    vocab_size: int = 128256
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    max_position_embeddings: int = 8192
    name: str = "llama-3-8b"

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads

    @property
    def q_size(self):
        return self.num_attention_heads * self.head_dim

    @property
    def kv_size(self):
        return self.num_key_value_heads * self.head_dim

    def packed_weight_bytes_per_layer(self):
        H, I = self.hidden_size, self.intermediate_size
        return ((self.q_size + 2 * self.kv_size) * H + H * H + 2 * I * H + H * I) // 2

    def algorithmic_bytes_per_forward(self):
        """packed weights + channel scales + fp16 lm_head (KV reads and activations excluded)."""
        H, I, L = self.hidden_size, self.intermediate_size, self.num_hidden_layers
        scales = 2 * (self.q_size + 2 * self.kv_size + H + 2 * I + H) * L
        return self.packed_weight_bytes_per_layer() * L + scales + 2 * self.vocab_size * H


CONFIGS = {
    "llama-3-8b": QuarotLlamaConfig(),
    "tinyllama-1.1b": QuarotLlamaConfig(2048, 5632, 32, 4, 22, 32000, 1e-5, 10000.0, 2048, "tinyllama-1.1b"),
    "llama-2-13b": QuarotLlamaConfig(5120, 13824, 40, 40, 40, 32000, 1e-5, 10000.0, 4096, "llama-2-13b"),
    "llama-3-70b": QuarotLlamaConfig(8192, 28672, 64, 8, 80, 128256, 1e-5, 500000.0, 8192, "llama-3-70b"),
}


@dataclass
class AttentionMetadata:
    """What vllm's FlashAttentionMetadata carries for this path (vllm/attention/backends/flash_attn.py:96-180)."""
    slot_mapping: torch.Tensor          # [T] i64
    block_tables: torch.Tensor          # [B, max_blocks] i32
    ctx_lens: torch.Tensor              # [B] i32: keys visible to the LAST query token of each sequence
    q_start: torch.Tensor               # [B+1] i32: query_start_loc
    max_q_len: int
    n_splits: int                       # context splits for the attention kernel (fixed per captured graph)


class Scratch:
    """Per-shape activation buffers, allocated once (the reference re-allocates ten of them every draft step,
    vllm/spec_decode/draft_model_runner.py:279-320; names kept)."""

    def __init__(self, cfg: QuarotLlamaConfig, T: int, n_seqs: int, max_q_len: int, n_splits: int, device,
                 logits_rows: Optional[int] = None):
        H, I = cfg.hidden_size, cfg.intermediate_size
        f16, i8 = torch.float16, torch.int8
        e = lambda *s, dtype=f16: torch.empty(*s, dtype=dtype, device=device)  # noqa: E731
        self.T = T
        self.hidden = e(T, H)
        self.normed = e(T, H)                                  # verify: fp16 LN output
        self.quantized_buffer_qkv = e(T, H // 2, dtype=i8)     # draft: int4 activations of width H
        self.quantized_buffer_mlp = e(T, I // 2, dtype=i8)
        self.scale_buffer = e(T)
        self.input_sum_buffer = e(T)
        self.act_buffer_qkv = e(T, cfg.q_size + 2 * cfg.kv_size)
        self.act_buffer_attn = e(T, cfg.q_size)
        self.act_buffer_had = e(T, cfg.q_size)
        self.act_buffer_output = e(T, H)
        self.act_buffer_gate_up = e(T, 2 * I)
        self.act_buffer_had_mlp = e(T, I)
        self.logits = e(T if logits_rows is None else logits_rows, cfg.vocab_size)
        # verify pass: raw fp32 K-slice sums of down_proj, finished inside the next norm (ops.w4a16_linear_partial)
        # (17..256 tokens: the M-tiled kernel's plan slices narrow layers up to eight ways)
        self.down_part = e(4 if T <= 16 else 8, T, H, dtype=torch.float32) if T <= 256 else None
        # verify pass at <= 16 tokens: fragment-major 16-row activation tiles between the producers and the W4A16 GEMMs
        # (ops.w4a16_act_layout_supported); rows past T are never read back into a result
        z = lambda k: torch.zeros(16, k, dtype=f16, device=device)  # noqa: E731
        self.xp_normed, self.xp_had, self.xp_had_mlp = (z(H), z(cfg.q_size), z(I)) if T <= 16 else (None, None, None)
        if 16 < T <= 32:   # two tiles each: the two-token-tile W4A16 streaming kernel (ops.w4a16_act_layout32_supported)
            self.xp_normed = torch.zeros(32, H, dtype=f16, device=device)
            self.xp_had = torch.zeros(32, cfg.q_size, dtype=f16, device=device)
            self.xp_had_mlp = torch.zeros(32, I, dtype=f16, device=device)
        self.tp_part = e(1, min(T, 32), H, dtype=torch.float32)   # TP verify pass (T <= 32): fp32 row-parallel partials
        self.had_part_amax = e(min(T, 16), 8, dtype=torch.float32)  # draft pass (T <= 4): partial row maxima of the spread head Hadamard
        # draft pass at 17..32 tokens: int32 K-slice sums of down_proj + the activation scales they were computed with
        self.down_ipart = e(2, T, H, dtype=torch.int32) if 16 < T <= 32 else None
        self.down_xs = e(T) if 16 < T <= 32 else None
        ws = ops.paged_attention_workspace_bytes(n_seqs * max_q_len, cfg.num_attention_heads, cfg.head_dim, n_splits)
        self.attn_ws = torch.zeros(ws, dtype=torch.uint8, device=device)   # ticket counters start at zero


class DecoderLayerWeights:
    def __init__(self, cfg: QuarotLlamaConfig, device):
        H, I = cfg.hidden_size, cfg.intermediate_size
        mk = lambda i, o: quarot_nn.Linear4bit(i, o, bias=False, device=device)  # noqa: E731
        self.qkv_proj = mk(H, cfg.q_size + 2 * cfg.kv_size)   # rows [q; k; v]   (fuse_qkv, quarot_llama.py:152-173)
        self.o_proj = mk(H, H)
        self.gate_up = mk(H, 2 * I)                            # rows [up; gate]  (fuse_gate_up, :301-314)
        self.down_proj = mk(I, H)

    def linears(self):
        return (self.qkv_proj, self.o_proj, self.gate_up, self.down_proj)


class QuarotLlamaForCausalLM:
    def __init__(self, cfg: QuarotLlamaConfig, device="cuda:0"):
        self.config = cfg
        self.device = torch.device(device)
        self.layers: List[DecoderLayerWeights] = [DecoderLayerWeights(cfg, device) for _ in range(cfg.num_hidden_layers)]
        self.embed_tokens = torch.zeros(cfg.vocab_size, cfg.hidden_size, dtype=torch.float16, device=device)
        self.lm_head = torch.zeros(cfg.vocab_size, cfg.hidden_size, dtype=torch.float16, device=device)
        had, self.had_K = hadamard_tables.get_hadK(cfg.intermediate_size)
        self.had_rem_dim = had.to(torch.float16).to(device) if had is not None else None
        hh, self.head_had_K = hadamard_tables.get_hadK(cfg.num_attention_heads)   # table factor for 12/20/28/40.. heads
        self.head_had = hh.to(torch.float16).to(device) if hh is not None else None
        self.head_had_scale = float(1.0 / torch.tensor(cfg.num_attention_heads).sqrt())      # hadamard.py:12
        self.mlp_had_scale = float(1.0 / torch.tensor(cfg.intermediate_size).sqrt())         # hadamard.py:13
        self.sm_scale = cfg.head_dim ** -0.5
        self.cos_sin_cache = make_cos_sin_cache(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta).to(device)
        # module-wise mirrors share the same buffers
        self.norm = quarot_nn.RMSNorm(cfg.hidden_size, cfg.rms_norm_eps)
        self.quantizer = quarot_nn.Quantizer()
        self.tp = None   # qspec_amd.parallel.TensorParallel for the verify pass (None = single GPU)

    # ------------------------------------------------------------------ weights
    @torch.no_grad()
    def init_synthetic(self, seed: int = 0, lm_head_std: float = 0.02):
        """SURVEY.md 8d synthetic model: int4 weights U{-8..7}, scales |N(0,1)|*0.01+1e-3, embed/lm_head N(0,0.02)."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        for layer in self.layers:
            for lin in layer.linears():
                n, kb = lin.weight.shape
                lin.weight.copy_(torch.randint(0, 256, (n, kb), generator=g, device=self.device, dtype=torch.int16)
                                 .to(torch.uint8).view(torch.int8))
                lin.weight_scales.copy_((torch.randn(n, 1, generator=g, device=self.device).abs() * 0.01 + 1e-3)
                                        .to(torch.float16))
        self.embed_tokens.copy_((torch.randn(self.embed_tokens.shape, generator=g, device=self.device) * 0.02)
                                .to(torch.float16))
        self.lm_head.copy_((torch.randn(self.lm_head.shape, generator=g, device=self.device) * lm_head_std)
                           .to(torch.float16))
        return self

    # M above which the fused GEMM epilogues (QKV+RoPE+KV write, gate_up+SiLU) give way to plain GEMM + separate kernels
    BIG_M = 64
    # verify pass: the fused-epilogue W4A16 forms are the 16-row streaming kernels; from 17 rows on the M-tiled kernel
    # (gemm_tiled.hip) + separate RoPE / SiLU launches is faster than the first-generation 2-D kernel that used to take
    # 17..64 rows (Llama-3-70B, bs = 8, T = 32: 51.2 -> 47.9 ms per cycle; Llama-3-8B bs = 8: 9.81 -> 9.56).  The
    # tensor-parallel shard views (K-sliced / channel-sharded streaming forms, T <= 32) keep the fused path.
    VERIFY_FUSE_MAX_M = int(__import__("os").environ.get("QSPEC_VERIFY_FUSE_MAX_M", "16"))
    # draft pass: LN in the GEMM prologue up to this many tokens (0 = always a separate LN launch; bit-identical).
    # Measured in the engine (round 2): bs = 4 fused 8.39 vs separate 8.43 ms per cycle; bs = 16 fused (hand-off form)
    # 14.30 vs separate 13.38 ms -- the standalone norm launch (3.4 us) beats the hand-off from 8 tokens on.
    FUSE_LN_MAX_M = int(__import__("os").environ.get("QSPEC_FUSE_LN_MAX_M", "4"))
    VERIFY_O_SLICES = int(__import__("os").environ.get("QSPEC_VERIFY_O_SLICES", "0"))   # dev knob, see the o_proj branch below
    FUSE_LN = True
    # draft pass, T <= 4, 32 heads of 128: head Hadamard spread over 8 workgroups per token + the quantiser in o_proj's prologue
    HADAMARD_QUANT_IN_OPROJ = __import__("os").environ.get("QSPEC_HQ_OPROJ", "1") != "0"

    # verify pass at <= 16 tokens (one GPU): the norm / head transform / MLP transform store their fp16 rows as the
    # FRAGMENT-MAJOR tile the W4A16 GEMMs' MFMA operands are loaded from (include/qspec_hip.h "activation layout"), which
    # removes the LDS regrouping pass from the head of four launches per layer.  Same bits either way.
    ACT_FRAGMENT_MAJOR = __import__("os").environ.get("QSPEC_ACT_FRAGMENT_MAJOR", "1") != "0"

    def _w4a16(self, x, lin, out, xp=False, tokens=None):
        # every M reads the packed int4 buffer: streaming kernel (M <= 16), M-tiled kernel (prefill-sized M)
        return ops.w4a16_linear(x, lin.weight, lin._scales(), out, xp=xp, tokens=tokens)

    def _add_norm_fp16(self, normed, hidden, delta, eps, xp=False):
        """hidden += delta; normed = LN(hidden).  delta: fp16 tensor, None, or ("partial", part, w_scale, S) = the raw
        K-slice sums of a long-K W4A16 down_proj, finished inside the norm kernel.  xp: normed is a fragment-major tile."""
        if isinstance(delta, tuple) and delta[0] == "ipartial":
            _, ipart, xs, w_scale, S = delta
            ops.add_rms_norm_ipartial(hidden, hidden, ipart, xs, w_scale, S, eps, out_f16=normed)
        elif isinstance(delta, tuple):
            _, part, w_scale, S = delta
            ops.add_rms_norm_fp16_partial(normed, hidden, hidden, part, w_scale, S, eps, xp=xp)
        else:
            ops.add_rms_norm_fp16(normed, hidden, hidden, delta, eps, xp=xp)

    def _fragment_major_ok(self, T, md, s):
        """Every producer and consumer of the verify pass has its fragment-major form at this shape."""
        cfg = self.config
        H, I, nh, hd = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.head_dim
        if not (self.ACT_FRAGMENT_MAJOR and s.xp_normed is not None and T <= 16 and self.MERGE_IN_HADAMARD
                and self.VERIFY_O_SLICES <= 1 and hd == 128 and md.n_splits <= 64):
            return False
        S = ops.w4a16_linear_partial_slices(T, H, I) if s.down_part is not None else 0
        down_k = I // S if 0 < S <= 4 else (I if S == 0 else 0)
        heads_ok = (nh == 32 if self.head_had_K == 1 else
                    ops.heads_hadamard_mix_merged_spread_supported(T, nh, hd, self.head_had_K))
        return bool(down_k and heads_ok and ops.w4a16_act_layout_supported(T, H)
                    and ops.w4a16_act_layout_supported(T, cfg.q_size) and ops.w4a16_act_layout_supported(T, down_k)
                    and ops.mlp_hadamard_act_layout_supported(T, I, self.had_K))

    XP32_DOWN_SLICES = __import__("os").environ.get("QSPEC_XP32_DOWN_SLICES", "1") != "0"
    TILED_PARTIAL_IN_NORM = __import__("os").environ.get("QSPEC_TILED_PARTIAL_IN_NORM", "1") != "0"
    MERGE_IN_HADAMARD = True   # False: the attention kernel merges its context splits itself (ticket + fences)
    DOWN_K_SLICES = __import__("os").environ.get("QSPEC_DOWN_K_SLICES", "1") != "0"   # draft down_proj at 17..32 tokens as K slices

    def _attention_hadamard(self, qkv, row, kc, vc, md, T, s, attn, q1, sc, had, xp=False):
        """Attention + head Hadamard (+ quant when q1 is given, else fp16 into `had`) (quarot_llama.py:213-238).
        For 32 / 64 heads of 128 the split merge of the attention kernel runs at the head of the Hadamard launch."""
        cfg = self.config
        nh = cfg.num_attention_heads
        B = md.ctx_lens.numel()
        # (other head sizes -- TinyLlama's 64 -- leave their split partials to the same launch when there are 32 heads)
        merged = (self.MERGE_IN_HADAMARD and md.n_splits <= 64 and self.head_had_K == 1
                  and ((cfg.head_dim == 128 and nh in (32, 64)) or (cfg.head_dim != 128 and cfg.head_dim % 8 == 0 and nh == 32)))
        # head count with a table factor (40 heads = had40): the split merge + transform spread over 8 workgroups per token
        mix_merged = (self.MERGE_IN_HADAMARD and self.head_had_K > 1 and md.n_splits <= 64
                      and ops.heads_hadamard_mix_merged_spread_supported(T, nh, cfg.head_dim, self.head_had_K))
        ops.paged_attention(qkv, row, kc, vc, md.block_tables, md.ctx_lens, md.q_start, T, md.max_q_len, nh,
                            self.sm_scale, md.n_splits, s.attn_ws, None if (merged or mix_merged) else attn)
        if self.head_had_K > 1:
            buf = had if had is not None else s.act_buffer_had[:T]
            if mix_merged:
                ops.heads_hadamard_mix_merged_spread(s.attn_ws, B * md.max_q_len, md.n_splits, T, nh, cfg.head_dim, self.head_had,
                                                     self.head_had_K, self.head_had_scale, buf, xp=xp)
            else:   # generic kernels: one workgroup per token
                ops.heads_hadamard_mix(attn.view(T, nh, cfg.head_dim), self.head_had, self.head_had_K, self.head_had_scale, buf)
            if q1 is not None:
                ops.fuse_sym_quant(buf, sc, q1)
        elif merged:
            ops.heads_hadamard_merged(s.attn_ws, B * md.max_q_len, md.n_splits, T, nh, cfg.head_dim,
                                      self.head_had_scale, out_f16=had, q=q1, scale=sc, xp=xp)
        elif q1 is not None:
            ops.heads_hadamard(attn, self.head_had_scale, q=q1, scale=sc, heads=nh)
        else:
            ops.heads_hadamard(attn, self.head_had_scale, out_f16=had, heads=nh)

    def weight_bytes(self):
        n = sum(lin.weight.numel() + lin.weight_scales.numel() * 2 for l in self.layers for lin in l.linears())
        return n + self.embed_tokens.numel() * 2 + self.lm_head.numel() * 2

    # ------------------------------------------------------------------ fused product path
    def forward(self, input_ids, positions, kv_caches, attn_metadata: AttentionMetadata, scratch: Scratch,
                w4a4: bool = False, embedded: bool = False, **kwargs):
        """LlamaModel.forward (quarot_llama.py:484-535) + final norm; returns the normed hidden states [T,H].
        embedded: scratch.hidden[:T] already holds embed_tokens[input_ids] (the engine's bookkeeping launches write it:
        ops.spec_prepare_draft / spec_advance_draft / spec_prepare_verify with embed=...)."""
        cfg, s, md = self.config, scratch, attn_metadata
        T = input_ids.numel()
        eps = cfg.rms_norm_eps
        hidden = s.hidden[:T]
        if not embedded:
            ops.embedding(input_ids, self.embed_tokens, hidden)
        delta = None
        qkv, attn, o, gu = s.act_buffer_qkv[:T], s.act_buffer_attn[:T], s.act_buffer_output[:T], s.act_buffer_gate_up[:T]
        q1, q3, sc = s.quantized_buffer_qkv[:T], s.quantized_buffer_mlp[:T], s.scale_buffer[:T]
        normed, had, had_mlp = s.normed[:T], s.act_buffer_had[:T], s.act_buffer_had_mlp[:T]
        row = cfg.q_size + 2 * cfg.kv_size
        tp_sharded = self.tp is not None and self.tp.world > 1 and getattr(self.tp, "shard_layers", True) and not w4a4
        # fused GEMM epilogues (decode-sized M)
        # (head size 64 -- TinyLlama -- has the fused qkv + RoPE + KV-write epilogue on the streaming kernels only)
        fuse_hd = cfg.head_dim == 128 or (cfg.head_dim == 64 and T <= 32 and
                                          ops.qkv_rope_linear_supported(w4a4, T, cfg.q_size + 2 * cfg.kv_size, cfg.hidden_size, 64))
        fuse = fuse_hd and (w4a4 or T <= (min(self.BIG_M, 32) if tp_sharded else self.VERIFY_FUSE_MAX_M))
        # tensor parallelism only on the verify pass at decode-sized M; the draft pass and prefill run replicated
        tp_on = tp_sharded and fuse and T <= 32
        act = s.act_buffer_had_mlp[:T]                            # silu(gate)*up, [T, I]
        nh, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        # draft pass at decode-sized M: residual add + LN + int4 quant run in the prologue of the qkv / gate_up GEMM
        # launches (gemm_stream.hip); the residual stream ping-pongs between two buffers
        ln_fused = (w4a4 and fuse and self.FUSE_LN and T <= self.FUSE_LN_MAX_M
                    and ops.ln_linear_s4s4_supported(T, row, cfg.hidden_size)
                    and ops.ln_linear_s4s4_supported(T, 2 * cfg.intermediate_size, cfg.hidden_size)
                    and ops.rowwise_scaled_linear_s4s4_residual_supported(T, cfg.hidden_size, cfg.hidden_size)
                    and ops.rowwise_scaled_linear_s4s4_residual_supported(T, cfg.hidden_size, cfg.intermediate_size))
        hq = (ln_fused and self.HADAMARD_QUANT_IN_OPROJ and self.MERGE_IN_HADAMARD and md.n_splits <= 64
              and (ops.heads_hadamard_merged_spread_supported(T, nh, hd) if self.head_had_K == 1 else
                   ops.heads_hadamard_mix_merged_spread_supported(T, nh, hd, self.head_had_K))
              and ops.rowwise_scaled_linear_s4s4_residual_hq_supported(T, cfg.hidden_size, cfg.q_size))
        had16 = s.act_buffer_had[:T]
        xp = not w4a4 and fuse and not tp_on and self._fragment_major_ok(T, md, s)
        if xp:
            normed_x, had_x = s.xp_normed, s.xp_had
        # verify pass at 17..32 tokens (one GPU): qkv / o_proj / gate_up on the two-token-tile W4A16 streaming kernel over two
        # fragment-major tiles (fused epilogues as at <= 16 tokens); down_proj stays on the M-tiled kernel (its K passes lose)
        xp32 = (not w4a4 and not fuse and not tp_sharded and 16 < T <= 32 and self.ACT_FRAGMENT_MAJOR and self.MERGE_IN_HADAMARD
                and hd == 128 and md.n_splits <= 64 and s.xp_normed is not None and s.xp_normed.shape[0] == 32
                and ops.w4a16_act_layout32_supported(T, row, cfg.hidden_size)
                and ops.w4a16_act_layout32_supported(T, cfg.hidden_size, cfg.q_size)
                and ops.w4a16_act_layout32_supported(T, 2 * cfg.intermediate_size, cfg.hidden_size)
                and (nh in (32, 64) if self.head_had_K == 1 else
                     ops.heads_hadamard_mix_merged_spread_supported(T, nh, hd, self.head_had_K)))
        # ... and down_proj as one-pass K slices of the same kernel across the workgroups, finished inside the next norm
        S32 = (ops.w4a16_linear_partial_slices_xp32(T, cfg.hidden_size, cfg.intermediate_size)
               if (xp32 and s.down_part is not None and self.XP32_DOWN_SLICES
                   and ops.mlp_hadamard_act_layout_supported(T, cfg.intermediate_size, self.had_K)) else 0)
        if xp32:
            normed_x, had_x = s.xp_normed, s.xp_had
        self.last_forward_form = "two-tile" if xp32 else ("fragment-major" if xp else ("fused" if fuse else "unfused"))   # (tests)
        # draft pass at 17..32 tokens: down_proj as K slices (0 / 1 = the plain launch)
        S_down = (ops.rowwise_scaled_linear_s4s4_partial_slices(T, cfg.hidden_size, cfg.intermediate_size)
                  if (w4a4 and fuse and not ln_fused and s.down_ipart is not None and self.DOWN_K_SLICES) else 0)
        for li, layer in enumerate(self.layers):
            kc, vc = kv_caches[li]
            qkv_w, qkv_s = layer.qkv_proj.weight, layer.qkv_proj._scales()
            gu_w, gu_s = layer.gate_up.weight, layer.gate_up._scales()
            if ln_fused:
                # 7 launches per layer; the residual stream is updated in the o_proj / down_proj epilogues
                # (hidden = residual + proj_out, :380,390), the norms read it in the qkv / gate_up prologues
                ops.ln_qkv_rope_linear(hidden, None, None, eps, qkv_w, qkv_s, qkv, positions, self.cos_sin_cache,
                                       kc, vc, md.slot_mapping, nh, nkv, hd)                       # :373-374,183-226
                if hq:
                    # merge + head Hadamard spread over 8 workgroups per token (fp16 + partial row maxima), the Quantizer
                    # of :235-238 in the o_proj launch's prologue: same bits as the two calls of the else branch
                    ops.paged_attention(qkv, row, kc, vc, md.block_tables, md.ctx_lens, md.q_start, T, md.max_q_len, nh,
                                        self.sm_scale, md.n_splits, s.attn_ws, None)
                    if self.head_had_K == 1:
                        ops.heads_hadamard_merged_spread(s.attn_ws, md.ctx_lens.numel() * md.max_q_len, md.n_splits, T, nh, hd,
                                                         self.head_had_scale, had16, s.had_part_amax[:T])
                    else:   # table-factor head count (Llama-2-13B: had40)
                        ops.heads_hadamard_mix_merged_spread(s.attn_ws, md.ctx_lens.numel() * md.max_q_len, md.n_splits, T, nh,
                                                             hd, self.head_had, self.head_had_K, self.head_had_scale, had16,
                                                             s.had_part_amax[:T])
                    ops.rowwise_scaled_linear_s4s4_residual_hq(had16, s.had_part_amax[:T], 1.0, layer.o_proj.weight,
                                                               layer.o_proj._scales(), hidden, hidden)
                else:
                    self._attention_hadamard(qkv, row, kc, vc, md, T, s, attn, q1, sc, None)        # :213-238
                    ops.rowwise_scaled_linear_s4s4_residual(q1, sc, layer.o_proj.weight, layer.o_proj._scales(), hidden, hidden)
                ops.ln_gate_up_silu_linear(hidden, None, None, eps, gu_w, gu_s, act)                # :380-388,266-284
                ops.mlp_hadamard(act, self.had_rem_dim, self.had_K, self.mlp_had_scale, q=q3, scale=sc)
                ops.rowwise_scaled_linear_s4s4_residual(q3, sc, layer.down_proj.weight, layer.down_proj._scales(), hidden, hidden)
                delta = None
                continue
            # input_layernorm (+ residual add of the previous MLP) -> qkv_proj -> rope -> kv write    :373-374,183-226
            if w4a4:
                if isinstance(delta, tuple):   # ("ipartial", ...): the previous down_proj's int32 K-slice sums
                    _, ipart, pxs, pws, S = delta
                    ops.add_rms_norm_ipartial(hidden, hidden, ipart, pxs, pws, S, eps, q=q1, scale=sc)
                else:
                    ops.add_rms_norm_i4(q1, sc, hidden, hidden, delta, eps)
                x, xs = q1, sc
            elif xp or xp32:
                self._add_norm_fp16(normed_x, hidden, delta, eps, xp=True)
                x, xs = normed_x, None
            else:
                self._add_norm_fp16(normed, hidden, delta, eps)
                x, xs = normed, None
            if xp32:
                ops.qkv_rope_linear_xp32(x, qkv_w, qkv_s, qkv, positions, self.cos_sin_cache, kc, vc, md.slot_mapping, nh, nkv, hd, T)
            elif fuse:
                ops.qkv_rope_linear(x, xs, qkv_w, qkv_s, qkv, positions, self.cos_sin_cache, kc, vc, md.slot_mapping,
                                    nh, nkv, hd, xp=xp, tokens=T)
            else:
                if w4a4:
                    ops.rowwise_scaled_linear_cutlass_s4s4_unified(q1, sc, qkv_w, qkv_s, None, qkv)
                else:
                    self._w4a16(normed, layer.qkv_proj, qkv)
                ops.rope_kv_write(positions, qkv, self.cos_sin_cache, kc, vc, md.slot_mapping, nh, nkv, hd)
            # attention -> heads hadamard (+ quant) -> o_proj -> residual + post_attention_layernorm     :231-243,380-387
            if w4a4:
                self._attention_hadamard(qkv, row, kc, vc, md, T, s, attn, q1, sc, None)
                ops.rowwise_scaled_linear_cutlass_s4s4_unified(q1, sc, layer.o_proj.weight, layer.o_proj._scales(), None, o)
                ops.add_rms_norm_i4(q1, sc, hidden, hidden, o, eps)
                x, xs = q1, sc
            elif xp:
                self._attention_hadamard(qkv, row, kc, vc, md, T, s, attn, None, None, had_x, xp=True)
                self._w4a16(had_x, layer.o_proj, o, xp=True, tokens=T)
                ops.add_rms_norm_fp16(normed_x, hidden, hidden, o, eps, xp=True)
                x, xs = normed_x, None
            elif xp32:
                self._attention_hadamard(qkv, row, kc, vc, md, T, s, attn, None, None, had_x, xp=True)
                ops.w4a16_linear_xp32(had_x, layer.o_proj.weight, layer.o_proj._scales(), o, T)
                ops.add_rms_norm_fp16(normed_x, hidden, hidden, o, eps, xp=True)
                x, xs = normed_x, None
            else:
                self._attention_hadamard(qkv, row, kc, vc, md, T, s, attn, None, None, had)
                if tp_on:   # row-parallel o_proj over this rank's K range of the shared buffer: raw fp32 sums,
                    # all-reduce in fp32, then scale + ONE fp16 rounding + residual add inside the norm (as on one GPU)
                    k0, k1 = self.tp.k_range(cfg.hidden_size)
                    part = s.tp_part[:, :T]
                    ops.w4a16_linear_ksliced_raw(had, layer.o_proj.weight, part[0], k0, k1)
                    self.tp.all_reduce(part)
                    ops.add_rms_norm_fp16_partial(normed, hidden, hidden, part, layer.o_proj._scales(), 1, eps)
                elif self.VERIFY_O_SLICES > 1 and fuse and s.down_part is not None and T <= 16:
                    # dev knob (QSPEC_VERIFY_O_SLICES = 2 / 4): o_proj as K slices whose raw sums the norm finishes, as
                    # down_proj's are -- measured, not the default: the slice order of the sum also
                    # breaks the bit identity with the module-wise path
                    S = self.VERIFY_O_SLICES
                    part = s.down_part.view(-1)[:S * T * cfg.hidden_size].view(S, T, cfg.hidden_size)
                    ops.w4a16_linear_partial(had, layer.o_proj.weight, part, S)
                    ops.add_rms_norm_fp16_partial(normed, hidden, hidden, part, layer.o_proj._scales(), S, eps)
                else:
                    # 17+ tokens: where the M-tiled kernel's plan slices K (narrow layers), the raw sums are finished inside
                    # the norm instead of by a finishing launch of their own (same expression, same bits)
                    S_o = (ops.w4a16_linear_partial_slices(T, cfg.hidden_size, cfg.q_size)
                           if (T > 16 and s.down_part is not None and self.TILED_PARTIAL_IN_NORM) else 0)
                    if 1 < S_o <= 8:
                        part = s.down_part.view(-1)[:S_o * T * cfg.hidden_size].view(S_o, T, cfg.hidden_size)
                        ops.w4a16_linear_partial(had, layer.o_proj.weight, part, S_o)
                        ops.add_rms_norm_fp16_partial(normed, hidden, hidden, part, layer.o_proj._scales(), S_o, eps)
                    else:
                        self._w4a16(had, layer.o_proj, o)
                        ops.add_rms_norm_fp16(normed, hidden, hidden, o, eps)
                x, xs = normed, None
            # gate_up -> silu*up -> online hadamard (+ quant) -> down_proj                              :266-299
            if xp32:
                ops.gate_up_silu_linear_xp32(x, gu_w, gu_s, act, T)
                if 1 < S32 <= 8:
                    ops.mlp_hadamard(act, self.had_rem_dim, self.had_K, self.mlp_had_scale, out_f16=s.xp_had_mlp, xp=True)
                    part = s.down_part.view(-1)[:S32 * T * cfg.hidden_size].view(S32, T, cfg.hidden_size)
                    ops.w4a16_linear_partial_xp32(s.xp_had_mlp, layer.down_proj.weight, part, S32, T)
                    delta = ("partial", part, layer.down_proj._scales(), S32)
                    continue
                had_mlp_in = s.act_buffer_gate_up.view(-1)[:T * cfg.intermediate_size].view(T, cfg.intermediate_size)
                ops.mlp_hadamard(act, self.had_rem_dim, self.had_K, self.mlp_had_scale, out_f16=had_mlp_in)
            elif fuse:
                if tp_on:   # column-parallel gate_up: own channels of [T, I], then all-gather of the channel ranges
                    c0, c1 = self.tp.channel_range(cfg.intermediate_size)
                    ops.gate_up_silu_linear_shard(x, gu_w, gu_s, act, c0, c1 - c0)
                    self.tp.all_gather_channels(act, cfg.intermediate_size)
                else:
                    ops.gate_up_silu_linear(x, xs, gu_w, gu_s, act, xp=xp, tokens=T)
                if w4a4:   # (K-sliced down_proj: its activation scales outlive the next norm's, so they get their own buffer)
                    ops.mlp_hadamard(act, self.had_rem_dim, self.had_K, self.mlp_had_scale, q=q3,
                                     scale=s.down_xs[:T] if S_down > 1 else sc)
                elif xp:
                    had_mlp_in = s.xp_had_mlp
                    ops.mlp_hadamard(act, self.had_rem_dim, self.had_K, self.mlp_had_scale, out_f16=had_mlp_in, xp=True)
                else:
                    had_mlp_in = s.act_buffer_gate_up.view(-1)[:T * cfg.intermediate_size].view(T, cfg.intermediate_size)
                    ops.mlp_hadamard(act, self.had_rem_dim, self.had_K, self.mlp_had_scale, out_f16=had_mlp_in)
            else:
                if w4a4:
                    ops.rowwise_scaled_linear_cutlass_s4s4_unified(q1, sc, gu_w, gu_s, None, gu)
                    ops.silu_mul_hadamard(gu, self.had_rem_dim, self.had_K, self.mlp_had_scale, q=q3, scale=sc)
                else:
                    self._w4a16(normed, layer.gate_up, gu)
                    ops.silu_mul_hadamard(gu, self.had_rem_dim, self.had_K, self.mlp_had_scale, out_f16=had_mlp)
                    had_mlp_in = had_mlp
            if w4a4:
                if S_down > 1:   # 17..32 tokens: K slices, int32 sums finished inside the next norm (ops.add_rms_norm_ipartial)
                    ipart = s.down_ipart.view(-1)[:S_down * T * cfg.hidden_size].view(S_down, T, cfg.hidden_size)
                    ops.rowwise_scaled_linear_s4s4_partial(q3, layer.down_proj.weight, ipart, S_down)
                    delta = ("ipartial", ipart, s.down_xs[:T], layer.down_proj._scales(), S_down)
                    continue
                ops.rowwise_scaled_linear_cutlass_s4s4_unified(q3, sc, layer.down_proj.weight, layer.down_proj._scales(), None, o)
            elif tp_on:     # row-parallel down_proj: raw fp32 sums, finished by the next norm
                k0, k1 = self.tp.k_range(cfg.intermediate_size)
                part = s.tp_part[:, :T]
                ops.w4a16_linear_ksliced_raw(had_mlp_in, layer.down_proj.weight, part[0], k0, k1)
                self.tp.all_reduce(part)
                delta = ("partial", part, layer.down_proj._scales(), 1)
                continue
            else:
                S = 0
                if s.down_part is not None and (fuse or (T > 16 and self.TILED_PARTIAL_IN_NORM)):
                    # long K at decode-sized M (streaming kernel) / narrow layer at 17+ tokens (M-tiled kernel's plan):
                    # K slices, finished by the next norm
                    S = ops.w4a16_linear_partial_slices(T, cfg.hidden_size, cfg.intermediate_size)
                if 0 < S <= (4 if T <= 16 else 8):
                    part = s.down_part.view(-1)[:S * T * cfg.hidden_size].view(S, T, cfg.hidden_size)
                    ops.w4a16_linear_partial(had_mlp_in, layer.down_proj.weight, part, S, xp=xp, tokens=T)
                    delta = ("partial", part, layer.down_proj._scales(), S)
                    continue
                self._w4a16(had_mlp_in, layer.down_proj, o, xp=xp, tokens=T)
            delta = o
        # final norm is always fp16, in both passes (self.norm(hidden_states) without kwargs, :533)
        self._add_norm_fp16(normed, hidden, delta, eps)
        return normed

    def compute_logits(self, hidden_states, scratch: Scratch, shard_vocab: bool = False):
        """LogitsProcessor with a plain nn.Linear lm_head (vllm/model_executor/layers/logits_processor.py:92-97).
        shard_vocab (verify pass under TP): this rank's vocab range + all-gather (logits_processor.py:104-107)."""
        T = hidden_states.shape[0]
        logits = scratch.logits[:T]
        if shard_vocab and self.tp is not None and self.tp.world > 1:
            v0, v1 = self.tp.vocab_range(self.config.vocab_size)
            local = self.tp._buf("vocab.local", (T, v1 - v0), torch.float16, self.device)
            ops.linear_f16(hidden_states, self.lm_head[v0:v1], local)
            return self.tp.all_gather_vocab(local, logits, self.config.vocab_size)
        ops.linear_f16(hidden_states, self.lm_head, logits)
        return logits

    def sample_greedy(self, hidden_states, scratch: Scratch, probs, token, shard_vocab: bool = False, logits_hook=None):
        """compute_logits + Sampler.forward (greedy, modify_greedy_probs = False: sampler.py:216-316): probs [T, V] fp32,
        token [T].  At decode-sized T on one GPU the lm_head launch and the softmax are fused (ops.lm_head_softmax_argmax);
        a vocab-parallel lm_head, a large T or a logits hook (bench only) take the two-step path.  Same bits."""
        T = hidden_states.shape[0]
        V, K = self.lm_head.shape
        sharded = shard_vocab and self.tp is not None and self.tp.world > 1
        if not sharded and logits_hook is None and ops.lm_head_softmax_argmax_supported(T, V, K):
            ops.lm_head_softmax_argmax(hidden_states, self.lm_head, scratch.logits[:T], probs, token)
            return
        logits = self.compute_logits(hidden_states, scratch, shard_vocab=shard_vocab)
        if logits_hook is not None:
            logits_hook(logits)
        ops.softmax_argmax(logits, probs, token)

    def sample(self, hidden_states, scratch: Scratch, probs, token, temperature, top_k, top_p, rng_state,
               shard_vocab: bool = False, logits_hook=None, exponential=None):
        """compute_logits + Sampler.forward for batches with non-greedy rows (sampler.py:216-316): temperature, top-k / top-p,
        softmax, multinomial by exponential noise; rows whose temperature is 0 take the argmax.  probs [T, V] fp32 (the
        PROCESSED distribution: what the rejection sampler must see), token [T]."""
        logits = self.compute_logits(hidden_states, scratch, shard_vocab=shard_vocab)
        if logits_hook is not None:
            logits_hook(logits)
        ops.sample_top_k_top_p(logits, probs, token, temperature, top_k, top_p, exponential=exponential, rng_state=rng_state)

    # ------------------------------------------------------------------ module-wise path (reference op order)
    def forward_modulewise(self, input_ids, positions, kv_caches, attn_metadata: AttentionMetadata, w4a4=False,
                           attn_override=None):
        """attn_override (tests): per layer an fp16 [T, q_size] tensor that REPLACES the attention output of that layer
        (teacher forcing of the one stage whose fp32 summation order the hardware fixes)."""
        cfg, md = self.config, attn_metadata
        kw = {"w4a4": w4a4}
        T = input_ids.numel()
        hidden = torch.empty(T, cfg.hidden_size, dtype=torch.float16, device=self.device)
        ops.embedding(input_ids, self.embed_tokens, hidden)
        head_had = quarot_nn.OnlineHadamard(cfg.num_attention_heads, device=self.device)
        mlp_had = quarot_nn.OnlineHadamard(cfg.intermediate_size, device=self.device)
        if self.had_rem_dim is not None:
            mlp_had.had_rem_dim = self.had_rem_dim
        ws = torch.zeros(ops.paged_attention_workspace_bytes(md.q_start.numel() * md.max_q_len, cfg.num_attention_heads,
                                                             cfg.head_dim, md.n_splits), dtype=torch.uint8, device=self.device)
        for li, layer in enumerate(self.layers):
            kc, vc = kv_caches[li]
            residual = hidden
            x = self.norm(hidden, **kw)
            qkv = layer.qkv_proj(x, **kw)
            q, k, v = qkv.split([cfg.q_size, cfg.kv_size, cfg.kv_size], dim=-1)
            ops.rotary_embedding(positions, q, k, cfg.head_dim, self.cos_sin_cache)
            ops.reshape_and_cache_flash(k.view(T, cfg.num_key_value_heads, cfg.head_dim),
                                        v.view(T, cfg.num_key_value_heads, cfg.head_dim), kc, vc, md.slot_mapping)
            attn = torch.empty(T, cfg.q_size, dtype=torch.float16, device=self.device)
            ops.paged_attention(qkv, qkv.shape[1], kc, vc, md.block_tables, md.ctx_lens, md.q_start, T, md.max_q_len,
                                cfg.num_attention_heads, self.sm_scale, md.n_splits, ws, attn)
            if attn_override is not None:
                attn.copy_(attn_override[li])
            a = attn.view(-1, cfg.num_attention_heads, cfg.head_dim)
            a = head_had(a.transpose(-1, -2).reshape(-1, cfg.num_attention_heads), **kw)          # :231
            a = a.view(-1, cfg.head_dim, cfg.num_attention_heads).transpose(-1, -2).reshape(T, cfg.hidden_size).contiguous()
            a = self.quantizer(a, **kw)
            hidden = ops_add(residual, layer.o_proj(a, **kw))                                      # :380
            residual = hidden
            x = self.norm(hidden, **kw)
            gu = layer.gate_up(x, **kw)
            g = ops.silu_mul(gu, torch.empty(T, cfg.intermediate_size, dtype=torch.float16, device=self.device))  # :279-284
            g = mlp_had(g, **kw).view(-1, cfg.intermediate_size)
            g = self.quantizer(g, **kw)
            hidden = ops_add(residual, layer.down_proj(g, **kw))                                   # :390
        return self.norm(hidden)


def ops_add(a, b):
    return a + b  # fp16 tensor add: one rounding per element, as in the reference


def make_cos_sin_cache(head_size: int, max_pos: int, base: float) -> torch.Tensor:
    """RotaryEmbedding._compute_cos_sin_cache (vllm/model_executor/layers/rotary_embedding.py), cast to fp16
    as quarot_llama.py:120 does; rope_scaling is ignored there and here."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_size, 2, dtype=torch.float32) / head_size))
    t = torch.arange(max_pos, dtype=torch.float32)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(torch.float16)
