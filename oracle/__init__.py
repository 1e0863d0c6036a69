"""CPU oracle for the QSpec draft/verify hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; ``qspec_amd`` never does (the product path fails
loudly when the HIP library is missing instead of falling back to this).

The arithmetic lives in ``qspec_oracle.c`` (plain C, built by ``oracle/Makefile``
into ``libqspec_oracle.so``); this module is the numpy binding plus the
numpy restatement of the rejection sampler / metrics / advance-step logic.
Each function cites the reference file:line it follows (paths relative to
the reference checkout).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libqspec_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "qspec_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libqspec_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _h(a):
    a = np.ascontiguousarray(a)
    assert a.dtype == np.float16, a.dtype
    return a


def _i8(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint8:
        a = a.view(np.int8)
    assert a.dtype == np.int8, a.dtype
    return a


c_int, c_float, c_i64 = ctypes.c_int, ctypes.c_float, ctypes.c_int64

# ---------------------------------------------------------------- primitives


def expf(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    lib().qo_expf(_p(x), _p(y), c_i64(x.size))
    return y


def pack_i4(q):
    """quarot/functional/quantization.py:42-49 (pack_i4)."""
    q = _i8(q)
    rows, cols = int(np.prod(q.shape[:-1])), q.shape[-1]
    out = np.empty(q.shape[:-1] + (cols // 2,), np.int8)
    lib().qo_pack_i4(_p(q), _p(out), c_i64(rows), c_i64(cols))
    return out


def unpack_i4(p):
    """quarot/functional/quantization.py:52-75 (unpack_i4)."""
    p = _i8(p)
    rows, cb = int(np.prod(p.shape[:-1])), p.shape[-1]
    out = np.empty(p.shape[:-1] + (cb * 2,), np.int8)
    lib().qo_unpack_i4(_p(p), _p(out), c_i64(rows), c_i64(cb * 2))
    return out


def ln_quant_i4(x, eps):
    """layernorm_kernels.cu:569-716 quant mode -> (q[T,H/2] int8, scale[T] f16, input_sum[T] f16)."""
    x = _h(x)
    T, H = x.shape
    q = np.empty((T, H // 2), np.int8)
    scale = np.empty((T,), np.float16)
    isum = np.empty((T,), np.float16)
    lib().qo_ln_quant_i4(_p(x), c_int(T), c_int(H), c_float(eps), _p(q), _p(scale), _p(isum))
    return q, scale, isum


def ln_fp16(x, eps):
    """layernorm_kernels.cu:927-957 fp16 mode."""
    x = _h(x)
    T, H = x.shape
    out = np.empty_like(x)
    lib().qo_ln_fp16(_p(x), c_int(T), c_int(H), c_float(eps), _p(out))
    return out


def rowabsmax_quant_i4(x, clip=1.0):
    """quant.cu:102-167 -> (q[T,K/2] int8, scale[T] f16)."""
    x = _h(x)
    T, K = x.shape
    q = np.empty((T, K // 2), np.int8)
    scale = np.empty((T,), np.float16)
    lib().qo_rowabsmax_quant_i4(_p(x), c_int(T), c_int(K), c_float(clip), _p(q), _p(scale))
    return q, scale


def fwht(x, scale):
    """fast_hadamard_transform_cuda.cu:124-198: rows of length N (power of two)."""
    x = _h(x)
    N = x.shape[-1]
    rows = x.size // N
    out = np.empty_like(x)
    lib().qo_fwht(_p(x), c_i64(rows), c_int(N), c_float(scale), _p(out))
    return out


def hadk_mix(y, hadK):
    """z[t] = hadK @ y[t]; y [T,K,M] f16, hadK [K,K] f16 (hadamard.py:104-108)."""
    y, hadK = _h(y), _h(hadK)
    T, K, M = y.shape
    z = np.empty_like(y)
    lib().qo_hadk_mix(_p(y), _p(hadK), c_int(T), c_int(K), c_int(M), _p(z))
    return z


def rsqrt_scale(n):
    """The fp32 value of ``1.0 / torch.tensor(n).sqrt()`` (quarot_nn/hadamard.py:12-13)."""
    return float(np.float32(1.0) / np.sqrt(np.float32(n)))


def heads_hadamard(attn_out, num_heads, scale=None, hadK=None, K=1):
    """o_proj online Hadamard over the head axis (quarot_llama.py:231-234):
    [T, heads*d] viewed [T,heads,d] -> transpose -> rows of `heads` -> OnlineHadamard(heads) -> transpose back.
    heads a power of two: FWHT * 1/sqrt(heads); a head count with a table factor (get_hadK(heads) = hadK, K > 1;
    40 heads = had40): matmul_hadU_cuda on those rows = FWHT over heads/K, then hadK mix (hadamard.py:94-124)."""
    x = _h(attn_out)
    T = x.shape[0]
    d = x.shape[1] // num_heads
    if scale is None:
        scale = rsqrt_scale(num_heads)
    xt = np.ascontiguousarray(x.reshape(T, num_heads, d).transpose(0, 2, 1)).reshape(T * d, num_heads)
    yt = fwht(xt, scale) if K == 1 else mlp_hadamard(xt, hadK, K, scale)
    return np.ascontiguousarray(yt.reshape(T, d, num_heads).transpose(0, 2, 1)).reshape(T, num_heads * d)


def mlp_hadamard(x, hadK, K, scale=None):
    """down_proj online Hadamard (quarot_nn/hadamard.py:36-39, quarot/functional/hadamard.py:94-124):
    view [T,K,n/K] -> FWHT over n/K (* 1/sqrt(n)) -> hadK mix over K -> [T,n]."""
    x = _h(x)
    T, n = x.shape
    if scale is None:
        scale = rsqrt_scale(n)
    y = fwht(x.reshape(T * K, n // K), scale)
    if K == 1:
        return y.reshape(T, n)
    return hadk_mix(y.reshape(T, K, n // K), hadK).reshape(T, n)


def silu_mul(gate_up, I):
    """quarot_llama.py:279-284: up = [:, :I], gate = [:, I:]; h(h(silu(gate)) * up)."""
    gu = _h(gate_up)
    T = gu.shape[0]
    out = np.empty((T, I), np.float16)
    lib().qo_silu_mul(_p(gu), c_int(T), c_int(I), _p(out))
    return out


def add_f16(a, b):
    a, b = _h(a), _h(b)
    out = np.empty_like(a)
    lib().qo_add_f16(_p(a), _p(b), _p(out), c_i64(a.size))
    return out


def gemm_w4a4(xq, xs, wq, ws, bias=None):
    """rowwise_scaled_linear_cutlass_unified.cuh:342-377 / ao test :64-84."""
    xq, wq, xs, ws = _i8(xq), _i8(wq), _h(xs), _h(ws).reshape(-1)
    M, Kb = xq.shape
    N = wq.shape[0]
    assert wq.shape[1] == Kb
    out = np.empty((M, N), np.float16)
    b = _h(bias) if bias is not None else None
    lib().qo_gemm_w4a4(_p(xq), _p(xs), _p(wq), _p(ws), _p(b) if b is not None else None, _p(out),
                       c_int(M), c_int(N), c_int(Kb * 2))
    return out


def gemm_w4a16(x, wq, ws, bias=None):
    """quarot_nn/linear.py:102-124 (bitblas.Matmul, W int4 per-channel scale, fp16 activations)."""
    x, wq, ws = _h(x), _i8(wq), _h(ws).reshape(-1)
    M, K = x.shape
    N = wq.shape[0]
    assert wq.shape[1] * 2 == K
    out = np.empty((M, N), np.float16)
    b = _h(bias) if bias is not None else None
    lib().qo_gemm_w4a16(_p(x), _p(wq), _p(ws), _p(b) if b is not None else None, _p(out),
                        c_int(M), c_int(N), c_int(K))
    return out


def gemm_w4a16_f32acc(x, wq, ws):
    """A second admissible W4A16 implementation (fp32 accumulate, interleaved partial sums): noise-floor probe."""
    x, wq, ws = _h(x), _i8(wq), _h(ws).reshape(-1)
    M, K = x.shape
    N = wq.shape[0]
    assert wq.shape[1] * 2 == K
    out = np.empty((M, N), np.float16)
    lib().qo_gemm_w4a16_f32acc(_p(x), _p(wq), _p(ws), _p(out), c_int(M), c_int(N), c_int(K))
    return out


def gemm_f16(x, w):
    """lm_head: x [M,K] @ w[N,K]^T -> f16 (logits_processor.py:92-97)."""
    x, w = _h(x), _h(w)
    M, K = x.shape
    N = w.shape[0]
    out = np.empty((M, N), np.float16)
    lib().qo_gemm_f16(_p(x), _p(w), _p(out), c_int(M), c_int(N), c_int(K))
    return out


def rope_neox(positions, q, k, cos_sin_cache, head_size):
    """csrc/pos_encoding_kernels.cu:10-35,71-92; in place on copies, returns (q,k)."""
    q, k, cs = _h(q).copy(), _h(k).copy(), _h(cos_sin_cache)
    pos = np.ascontiguousarray(positions, dtype=np.int64)
    T = q.shape[0]
    nq, nk = q.shape[1] // head_size, k.shape[1] // head_size
    lib().qo_rope_neox(_p(pos), _p(q), _p(k), _p(cs), c_int(T), c_int(nq), c_int(nk), c_int(head_size),
                       c_int(cs.shape[1]), c_i64(q.shape[1]), c_i64(k.shape[1]))
    return q, k


def make_cos_sin_cache(head_size, max_pos, base):
    """vllm rotary_embedding.py:136-150 RotaryEmbedding._compute_cos_sin_cache, cast to fp16
    (quarot_llama.py:112-120: plain rotary, rope_scaling ignored).

    Pinned against the reference's own table (tests/golden/rope_cache_softmax.npz): the table is fp32 `pow`, `cos`,
    `sin` of angles up to max_pos radians, and a numpy restatement differed from the reference's torch-CPU result by up
    to 8 fp16 ulps in 0.8 % of the entries (numpy's and torch's fp32 `pow` differ by an ulp, the angle multiplies that
    by the position).  The restatement therefore uses the same torch CPU operators in the same order: bit-identical."""
    import torch
    inv_freq = 1.0 / (base ** (torch.arange(0, head_size, 2, dtype=torch.float) / head_size))
    t = torch.arange(max_pos, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(torch.float16).numpy()


def reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping):
    """csrc/cache_kernels.cu:207-247: caches [num_blocks, block_size, n_kv, d], in place."""
    nb, bs, nkv, d = key_cache.shape
    kc = key_cache.reshape(nb * bs, nkv * d)
    vc = value_cache.reshape(nb * bs, nkv * d)
    for t, s in enumerate(np.asarray(slot_mapping)):
        if s < 0:
            continue
        kc[s] = key[t].reshape(-1)
        vc[s] = value[t].reshape(-1)


def paged_attention(q, key_cache, value_cache, block_tables, ctx_lens, q_start, sm_scale):
    """Causal varlen attention over the paged fp16 cache (flash_attn.py:741-830 contract)."""
    q = _h(q)
    kc, vc = _h(key_cache), _h(value_cache)
    nb, bs, nkv, d = kc.shape
    T = q.shape[0]
    nq = q.shape[1] // d
    bt = np.ascontiguousarray(block_tables, dtype=np.int32)
    cl = np.ascontiguousarray(ctx_lens, dtype=np.int32)
    qs = np.ascontiguousarray(q_start, dtype=np.int32)
    out = np.empty((T, nq * d), np.float16)
    lib().qo_paged_attention(_p(q), _p(kc), _p(vc), _p(bt), c_int(bt.shape[1]), _p(cl), _p(qs), c_int(len(cl)),
                             c_int(nq), c_int(nkv), c_int(d), c_int(bs), c_float(sm_scale), _p(out))
    return out


def softmax_argmax(logits):
    """sampler.py:270-287 greedy with modify_greedy_probs=False -> (probs f32 [T,V], token i64 [T])."""
    l = _h(logits)
    T, V = l.shape
    probs = np.empty((T, V), np.float32)
    tok = np.empty((T,), np.int64)
    lib().qo_softmax_argmax(_p(l), c_int(T), c_int(V), _p(probs), _p(tok))
    return probs, tok


# --------------------------------------------------- rejection sampler (numpy)

FLT_TINY = np.finfo(np.float32).tiny


def rejection_sample(target_with_bonus_probs, bonus_token_ids, draft_probs, draft_token_ids, uniform, exponential):
    """RejectionSampler.forward with the random draws injected
    (vllm/model_executor/layers/rejection_sampler.py:60-154, _get_accepted :252-299,
    _get_recovered_probs :301-349, _multinomial :374-399) and _create_output
    (spec_decode_base_sampler.py:69-131).

    uniform [B,k] f32 stands for ``torch.rand``; exponential [B,k,V] f32 for
    ``q.exponential_(1.0)``.  Returns (output [B,k+1] i64, accepted [B,k] bool,
    recovered [B,k] i64, counters (accepted, emitted, draft))."""
    q = np.asarray(target_with_bonus_probs, np.float32)[:, :-1]
    p = np.asarray(draft_probs, np.float32)
    ids = np.asarray(draft_token_ids, np.int64)
    B, k, V = p.shape
    bi = np.arange(B)[:, None]
    ki = np.arange(k)[None, :]
    sel_p = p[bi, ki, ids]
    sel_q = q[bi, ki, ids]
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.minimum(sel_q / sel_p, np.float32(1.0)).astype(np.float32)
    accepted = np.asarray(uniform, np.float32) < ratio
    f = np.maximum(q - p, np.float32(FLT_TINY)).astype(np.float32)
    # torch.sum over V in fp32 has no specified order; the oracle sums in fp64 and rounds once.
    s = f.sum(axis=-1, dtype=np.float64).astype(np.float32)
    rec = (f / s[..., None]).astype(np.float32)
    recovered = np.argmax((rec / np.asarray(exponential, np.float32)).astype(np.float32), axis=-1).astype(np.int64)
    out, counters = create_output(accepted, recovered, ids, np.asarray(bonus_token_ids, np.int64).reshape(B))
    return out, accepted, recovered, counters


def sample_top_k_top_p(logits, temperature, top_k, top_p, exponential):
    """Sampler.forward for non-greedy rows (vllm/model_executor/layers/sampler.py:216-316) with the exponential draws injected:
    l = f32(logits) / T; _apply_top_k_top_p (:387-413) op for op -- ascending sort, top-k threshold = sorted[V - k] (ties kept),
    softmax of the rest, ascending fp32 cumsum (sequential, as torch on CPU), mask where cumsum <= 1 - p, never the last --;
    probs = softmax(masked l); token = argmax(probs / E) (_multinomial :585-604), argmax(probs) for rows with T < 1e-5.
    The final softmax uses the oracle's own exp (qexpf) and an fp64 denominator like softmax_argmax.
    Returns (probs [T,V] f32, token [T] i64, keep mask [T,V] bool)."""
    lg = np.asarray(logits, np.float16).astype(np.float32)
    Tn, V = lg.shape
    temperature = np.ones(Tn, np.float32) if temperature is None else np.asarray(temperature, np.float32)
    top_k = np.full(Tn, -1, np.int64) if top_k is None else np.asarray(top_k, np.int64)
    top_p = np.ones(Tn, np.float32) if top_p is None else np.asarray(top_p, np.float32)
    probs = np.zeros((Tn, V), np.float32)
    keep = np.ones((Tn, V), bool)
    tok = np.zeros(Tn, np.int64)
    for t in range(Tn):
        greedy = temperature[t] < 1e-5
        l = (lg[t] / (np.float32(1.0) if greedy else temperature[t])).astype(np.float32)
        order = np.argsort(l, kind="stable")
        ls = l[order].copy()
        k = int(top_k[t])
        if k <= 0 or k > V:
            k = V
        ls[ls < ls[V - k]] = -np.inf
        e = np.exp((ls - ls[-1]).astype(np.float64))
        ps = (e / e.sum()).astype(np.float32)
        cs = np.cumsum(ps, dtype=np.float32)
        m = cs <= np.float32(1.0) - top_p[t]
        m[-1] = False
        ls[m] = -np.inf
        keep[t, order] = np.isfinite(ls)
        mx = l[keep[t]].max()
        ex = np.where(keep[t], expf((l - mx).astype(np.float32)), np.float32(0.0)).astype(np.float32)
        z = np.float32(ex.astype(np.float64).sum())
        probs[t] = (ex / z).astype(np.float32)
        tok[t] = int(np.argmax(probs[t])) if greedy else int(np.argmax((probs[t] / np.asarray(exponential, np.float32)[t]).astype(np.float32)))
    return probs, tok, keep


def typical_acceptance_sample(target_with_bonus_probs, bonus_token_ids, draft_token_ids, posterior_threshold, posterior_alpha):
    """TypicalAcceptanceSampler.forward (vllm/model_executor/layers/typical_acceptance_sampler.py:37-172):
    accepted = q[x] > min(posterior_threshold, posterior_alpha * exp(-H)), H = -sum_v q log(q + 1e-5) (fp32 terms as the
    reference forms them; torch.sum has no specified order: summed in fp64, rounded once), replacement = argmax_v q, then
    _create_output.  Returns (output [B,k+1] i64, accepted [B,k] bool, recovered [B,k] i64, counters, entropy [B,k] f32)."""
    q = np.asarray(target_with_bonus_probs, np.float32)[:, :-1]
    ids = np.asarray(draft_token_ids, np.int64)
    B, k, V = q.shape
    cand = q[np.arange(B)[:, None], np.arange(k)[None, :], ids]
    terms = (q * np.log(q + np.float32(1e-5), dtype=np.float32)).astype(np.float32)
    H = (-terms.sum(axis=-1, dtype=np.float64)).astype(np.float32)
    thr = np.minimum(np.float32(posterior_threshold), (expf(-H) * np.float32(posterior_alpha)).astype(np.float32))
    accepted = cand > thr
    recovered = np.argmax(q, axis=-1).astype(np.int64)
    out, counters = create_output(accepted, recovered, ids, np.asarray(bonus_token_ids, np.int64).reshape(B))
    return out, accepted, recovered, counters, H


def create_output(accepted, substitute_token_ids, draft_token_ids, bonus_token_ids):
    """spec_decode_base_sampler.py:69-131."""
    accepted = np.asarray(accepted, bool)
    B, k = accepted.shape
    rej = ~accepted
    limits = np.where(rej.any(1), rej.argmax(1), k)
    idx = np.arange(k)[None, :]
    acc_mask = idx < limits[:, None]
    after = idx == limits[:, None]
    out = -np.ones((B, k + 1), np.int64)
    out[:, :k] = np.where(acc_mask, draft_token_ids, -1)
    out[:, -1] = np.where(out[:, k - 1] != -1, bonus_token_ids, -1)
    out[:, :k] = out[:, :k] * (~after) + substitute_token_ids * after
    counters = (int(accepted.sum()), int((out != -1).sum()), B * k)
    return out, counters


def spec_metrics(num_accepted, num_emitted, num_draft, k):
    """vllm/spec_decode/metrics.py:164-188 (draft_acceptance_rate, system_efficiency)."""
    rate = num_accepted / num_draft if num_draft > 0 else float("nan")
    max_emitted = (num_draft // k) * (k + 1) if k > 0 else 0
    eff = num_emitted / max_emitted if max_emitted > 0 else float("nan")
    return rate, eff


def advance_step(input_tokens, sampled, positions, seq_lens, slot_mapping, block_tables, block_size):
    """csrc/prepare_inputs/advance_step.cu:14-64 (no padding), in place."""
    n = len(seq_lens)
    for i in range(n):
        input_tokens[i] = sampled[i]
        nsl = seq_lens[i] + 1
        seq_lens[i] = nsl
        positions[i] = nsl - 1
        slot_mapping[i] = block_tables[i, (nsl - 1) // block_size] * block_size + (nsl - 1) % block_size


def count_div3_mismatches(s_lo: int, s_hi: int) -> int:
    """Exhaustive check of the HIP quantisers' three-instruction h(x / s) (common.cuh:div3_h) against the fp16
    rounding of the correctly rounded fp32 quotient, for scales with fp16 bit patterns in [s_lo, s_hi) and every
    non-negative finite fp16 x."""
    f = lib().qo_count_div3_mismatches
    f.restype = ctypes.c_longlong
    return int(f(c_int(s_lo), c_int(s_hi)))
