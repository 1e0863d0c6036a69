"""Pin the CPU oracle against the golden vectors produced by running the
importable parts of the reference (tests/golden/make_golden.py) and against the
reference tests' own known-answer constructions.  CPU only."""
import os

import numpy as np
import pytest


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_pack_unpack_matches_reference(oracle, golden_dir):
    g = _load(golden_dir, "pack_i4.npz")
    assert np.array_equal(oracle.pack_i4(g["q"]).view(np.uint8), g["packed"])
    assert np.array_equal(oracle.unpack_i4(g["packed"]), g["unpacked"].astype(np.int8))
    # every nibble value, both positions
    q = np.array([[a, b] for a in range(-8, 8) for b in range(-8, 8)], np.int8)
    assert np.array_equal(oracle.unpack_i4(oracle.pack_i4(q)), q)


def test_hadK_tables_are_hadamard(golden_dir):
    g = _load(golden_dir, "hadamard.npz")
    for K in (12, 20, 28, 36, 40, 52, 60, 108):
        h = g[f"had{K}"].astype(np.int64)
        assert set(np.unique(h)) == {-1, 1}
        assert np.array_equal(h @ h.T, K * np.eye(K, dtype=np.int64))


@pytest.mark.parametrize("n", [32, 512])
def test_fwht_matches_reference_matmul_hadU(oracle, golden_dir, n):
    """fast-hadamard-transform's own test compares with scipy.linalg.hadamard at atol 5e-3 (fp16)
    (third-party/fast-hadamard-transform/tests/test_fast_hadamard_transform.py:13-38)."""
    from scipy.linalg import hadamard
    g = _load(golden_dir, "hadamard.npz")
    x, y = g[f"x_{n}"], g[f"y_{n}"]
    out = oracle.fwht(x.astype(np.float16), oracle.rsqrt_scale(n)).astype(np.float64)
    assert np.allclose(out, y, atol=5e-3, rtol=0)
    assert np.allclose(x @ hadamard(n).T / np.sqrt(n), y, atol=1e-9)  # Sylvester order


@pytest.mark.parametrize("n,K", [(14336, 28), (28672, 28), (13824, 108)])
def test_mlp_hadamard_matches_reference_matmul_hadU(oracle, golden_dir, n, K):
    """(hadK (x) H_{n/K}) / sqrt(n) with element index k*(n/K)+j, as quarot/functional/hadamard.py:59-80."""
    g = _load(golden_dir, "hadamard.npz")
    x, y = g[f"x_{n}"], g[f"y_{n}"]
    hadK = g[f"had{K}"].astype(np.float16)
    out = oracle.mlp_hadamard(x.astype(np.float16), hadK, K).astype(np.float64)
    assert np.allclose(out, y, atol=5e-3, rtol=0), np.abs(out - y).max()


def test_heads_hadamard_is_transpose_fwht_transpose(oracle):
    from scipy.linalg import hadamard
    rng = np.random.default_rng(0)
    T, heads, d = 3, 32, 128
    x = rng.standard_normal((T, heads * d)).astype(np.float16)
    out = oracle.heads_hadamard(x, heads).astype(np.float64)
    ref = np.einsum("gh,thd->tgd", hadamard(heads) / np.sqrt(heads), x.reshape(T, heads, d).astype(np.float64))
    assert np.allclose(out, ref.reshape(T, -1), atol=5e-3, rtol=0)


def test_w4a4_gemm_bit_exact_with_reference_formula(oracle, golden_dir):
    """third-party/ao/test/test_rowwise_scaled_linear_cutlass.py:64-84: the fp32 formula is exact for
    int4 products, so the oracle must reproduce it bit for bit."""
    g = _load(golden_dir, "w4a4_gemm.npz")
    idx = 0
    while f"c{idx}_xq" in g:
        xq, wq, xs, ws = g[f"c{idx}_xq"], g[f"c{idx}_wq"], g[f"c{idx}_xs"], g[f"c{idx}_ws"]
        out = oracle.gemm_w4a4(xq, xs, wq, ws)
        assert np.array_equal(out.view(np.uint16), g[f"c{idx}_ref_n"].view(np.uint16)), idx
        out = oracle.gemm_w4a4(xq, xs, wq, ws, g[f"c{idx}_bias"])
        assert np.array_equal(out.view(np.uint16), g[f"c{idx}_ref_b"].view(np.uint16)), idx
        idx += 1
    assert idx == 6


def test_w4a16_matches_dequantised_matmul(oracle):
    """Second statements of the W4A16 semantics in the reference: unpack * scale then fp16 matmul
    (quarot_nn/linear.py:111-119) and the fp32-accumulate Triton kernel (quarot_nn/qspec_gemm.py:20-88)."""
    rng = np.random.default_rng(1)
    M, N, K = 5, 64, 256
    x = rng.standard_normal((M, K)).astype(np.float16)
    w = rng.integers(-8, 8, (N, K)).astype(np.int8)
    ws = (rng.random(N) * 0.01 + 0.001).astype(np.float16)
    out = oracle.gemm_w4a16(x, oracle.pack_i4(w), ws).astype(np.float64)
    ref = (x.astype(np.float64) @ w.astype(np.float64).T) * ws.astype(np.float64)
    assert np.allclose(out, ref, rtol=1e-3, atol=1e-4)


def test_w4a16_second_implementation_meets_the_same_bar(oracle):
    """The fp32-accumulate variant (noise-floor probe of the full-depth GPU test) is itself within 1e-3 of the fp64 one
    at the decoder's K sizes."""
    rng = np.random.default_rng(2)
    for M, N, K in [(4, 96, 4096), (3, 64, 14336)]:
        x = rng.standard_normal((M, K)).astype(np.float16)
        w = oracle.pack_i4(rng.integers(-8, 8, (N, K)).astype(np.int8))
        ws = (rng.random(N) * 0.01 + 0.001).astype(np.float16)
        a = oracle.gemm_w4a16(x, w, ws).astype(np.float64)
        b = oracle.gemm_w4a16_f32acc(x, w, ws).astype(np.float64)
        assert (np.abs(a - b) <= 1e-3 * np.maximum(1.0, np.abs(a))).all()


def test_rejection_sampler_matches_reference_run(oracle, golden_dir):
    """The reference RejectionSampler itself, run on CPU with recorded uniform / exponential draws."""
    g = _load(golden_dir, "rejection.npz")
    n = 0
    for idx in range(5):
        for flavour in ("random", "agree", "onehot"):
            key = f"c{idx}_{flavour}"
            out, accepted, recovered, counters = oracle.rejection_sample(
                g[key + "_tq"], g[key + "_bonus"], g[key + "_dp"], g[key + "_ids"], g[key + "_U"], g[key + "_E"])
            assert np.array_equal(out, g[key + "_out"]), key
            assert list(counters) == list(g[key + "_counters"]), key
            n += 1
    assert n == 15


@pytest.mark.parametrize("name", ["all", "none", "some"])
def test_create_output_known_answers(oracle, golden_dir, name):
    """tests/samplers/test_rejection_sampler.py:48-127 (exact output layout)."""
    g = _load(golden_dir, "rejection.npz")
    acc, rec, ids, bonus = (g[f"co_{name}_{k}"] for k in ("accepted", "rec", "ids", "bonus"))
    out, _ = oracle.create_output(acc, rec, ids, bonus.reshape(-1))
    assert np.array_equal(out, g[f"co_{name}_out"])
    if name == "all":
        assert np.array_equal(out[:, :-1], ids) and np.array_equal(out[:, -1], bonus.reshape(-1))
    if name == "none":
        assert np.array_equal(out[:, 0], rec[:, 0]) and (out[:, 1:] == -1).all()


def test_spec_metrics_formulas(oracle):
    """vllm/spec_decode/metrics.py:164-188 on the numbers of the reference's screenshot:
    3931 accepted / 4092 draft / 5161 emitted at k=3 -> 0.961 / 0.946."""
    rate, eff = oracle.spec_metrics(3931, 5161, 4092, 3)
    assert abs(rate - 0.9607) < 1e-3 and abs(eff - 0.9460) < 1e-3


# ----------------------------------------------------------------- edge rows of the unpinned restatements

def test_ln_quant_edge_rows(oracle):
    H = 4096
    x = np.zeros((4, H), np.float16)
    x[1] = 3.0                                  # constant row: var = 0, values 0 -> amax floor 1e-6
    x[2, ::2], x[2, 1::2] = 1.0, -1.0           # +-1: normalised +-1 -> q = +-7
    x[3, 0] = 100.0                             # one outlier
    q, scale, isum = oracle.ln_quant_i4(x, 1e-5)
    u = oracle.unpack_i4(q)
    assert (u[0] == 0).all() and (u[1] == 0).all()
    assert scale[0] == np.float16(np.float32(np.float16(1e-6)) / 7) == scale[1]
    assert set(np.unique(u[2])) == {-7, 7}
    assert u[3, 0] == 7 and np.abs(u[3, 1:]).max() == 0
    out = oracle.ln_fp16(x, 1e-5)
    assert np.allclose(out[2].astype(np.float32), x[2].astype(np.float32), atol=1e-3)


def test_rowabsmax_edge_rows(oracle):
    K = 256
    x = np.zeros((3, K), np.float16)            # row 0 all zero -> scale 0 -> 0/0 = NaN -> 0
    x[1] = np.linspace(-7, 7, K).astype(np.float16)
    x[2, :4] = [3.5, 2.5, -2.5, -3.5]           # exact .5 ties at scale 1 -> round half to even
    x[2, 4] = 7.0
    q, scale = oracle.rowabsmax_quant_i4(x, 1.0)
    u = oracle.unpack_i4(q)
    assert (u[0] == 0).all() and scale[0] == 0
    assert scale[1] == np.float16(1.0) and u[1, 0] == -7 and u[1, -1] == 7
    assert list(u[2, :5]) == [4, 2, -2, -4, 7]


def test_three_op_fp16_division_is_exact_for_every_fp16_pair(oracle):
    """The HIP quantisers compute h(x / scale) as q0 = x*r, q1 = q0 + (x - q0*s)*r with r = 1/s (common.cuh:div3_h).
    For fp16-valued x and s that is the fp16 rounding of the IEEE quotient for EVERY pair: all 31743 positive finite
    scales x all 31744 non-negative finite x (the expression is odd in x), 10^9 pairs."""
    assert oracle.count_div3_mismatches(1, 0x7C00) == 0


@pytest.mark.parametrize("heads", [40, 80])
def test_heads_hadamard_with_table_factor_matches_reference_matmul_hadU(oracle, golden_dir, heads):
    """Head counts whose get_hadK factor is a table (Llama-2-13B: 40 heads = had40; 80 = had40 (x) H2): the head
    transform is matmul_hadU on rows of `heads` between the two transposes (quarot_llama.py:231-234); the reference's
    own matmul_hadU was run on these rows (tests/golden/make_golden.py)."""
    g = _load(golden_dir, "hadamard.npz")
    x, y = g[f"x_{heads}"], g[f"y_{heads}"]          # [3, heads]: three (token, d) columns
    hadK = g["had40"].astype(np.float16)
    d = 1
    attn = x.astype(np.float16).reshape(3, heads * d)   # [T, heads, d = 1]
    out = oracle.heads_hadamard(attn, heads, None, hadK, 40).astype(np.float64)
    assert np.allclose(out, y, atol=5e-3, rtol=0), np.abs(out - y).max()


# ----------------------------------------------------------------- f1 / a11: attention, RoPE, KV write, greedy softmax
# Fixtures: the reference's own pure-torch formulas run on CPU by tests/golden/make_golden.py
# (tests/kernels/test_flash_attn.py:19-75, vllm/model_executor/layers/rotary_embedding.py:47-70,136-150,201-229,
#  tests/kernels/test_cache.py:295-304, vllm/model_executor/layers/sampler.py:278-287).

def attention_case(g, idx):
    """One case of attention.npz as the arguments of the paged-attention contract: q [T, nq*d], caches, block tables,
    context lengths, query starts, softmax scale, and the two reference outputs reshaped to [T, nq*d]."""
    key = f"c{idx}_"
    q = g[key + "q"]
    T, nq, d = q.shape
    q_lens = g[key + "query_lens"]
    q_start = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    return dict(q=q.reshape(T, nq * d), key_cache=g[key + "key_cache"], value_cache=g[key + "value_cache"],
                block_tables=g[key + "block_tables"], ctx_lens=g[key + "kv_lens"], q_start=q_start,
                scale=float(g[key + "scale"]), ref16=g[key + "ref16"].reshape(T, nq * d),
                ref32=g[key + "ref32"].reshape(T, nq * d), n_cases=3)


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_paged_attention_matches_reference_ref_paged_attn(oracle, golden_dir, idx):
    """Paged KV through block tables, GQA head mapping, causal mask of a varlen batch with the queries at the END of
    each context: the oracle against ref_paged_attn.  <= 1e-3 against the function on fp32-held values; against the
    fp16 call (S and P rounded to fp16 inside it) the reference's own bar for flash-attn, atol 2e-2 / rtol 1e-2
    (test_flash_attn.py:155-156)."""
    c = attention_case(_load(golden_dir, "attention.npz"), idx)
    out = oracle.paged_attention(c["q"], c["key_cache"], c["value_cache"], c["block_tables"], c["ctx_lens"],
                                 c["q_start"], c["scale"]).astype(np.float64)
    err = np.abs(out - c["ref32"])
    assert err.max() <= 1e-3, err.max()          # (includes the oracle's one rounding to fp16: |out| < 2 here)
    assert np.allclose(out, c["ref16"].astype(np.float64), atol=2e-2, rtol=1e-2)


@pytest.mark.parametrize("idx", [0, 1])
def test_rope_matches_reference_forward_native_bit_exact(oracle, golden_dir, idx):
    """cos/sin table = RotaryEmbedding._compute_cos_sin_cache cast to fp16, rotation = forward_native on fp16 tensors
    (every operator rounds to fp16, as the CUDA kernel's scalar_t arithmetic does): bit for bit."""
    g = _load(golden_dir, "rope_cache_softmax.npz")
    key = f"rope{idx}_"
    d, max_pos, nq, nkv = (int(v) for v in g[key + "cfg"])
    cs = oracle.make_cos_sin_cache(d, max_pos, float(g[key + "base"]))
    rows = g[key + "cache_rows"]
    assert np.array_equal(cs[rows].view(np.uint16), g[key + "cache"].view(np.uint16))   # the table, bit for bit
    from qspec_amd.model import make_cos_sin_cache               # the product's table: the same bits
    assert np.array_equal(make_cos_sin_cache(d, max_pos, float(g[key + "base"])).numpy()[rows].view(np.uint16),
                          g[key + "cache"].view(np.uint16))
    full = cs
    q, k = oracle.rope_neox(g[key + "pos"], g[key + "q"], g[key + "k"], full, d)
    assert np.array_equal(q.view(np.uint16), g[key + "q_out"].view(np.uint16))
    assert np.array_equal(k.view(np.uint16), g[key + "k_out"].view(np.uint16))


def test_reshape_and_cache_flash_matches_reference_loop(oracle, golden_dir):
    g = _load(golden_dir, "rope_cache_softmax.npz")
    kc, vc = g["cache_key_cache_in"].copy(), g["cache_value_cache_in"].copy()
    oracle.reshape_and_cache_flash(g["cache_key"], g["cache_value"], kc, vc, g["cache_slot_mapping"])
    assert np.array_equal(kc.view(np.uint16), g["cache_key_cache_out"].view(np.uint16))
    assert np.array_equal(vc.view(np.uint16), g["cache_value_cache_out"].view(np.uint16))


def test_greedy_softmax_argmax_matches_torch(oracle, golden_dir):
    """probs vs torch.softmax(fp32) <= 1e-3 relative (measured: a few fp32 ulps -- `qexpf` is within 1.4 ulp of expf
    and the denominator is summed in fp64), token == argmax(log_softmax) on every row incl. an exact tie (first
    index), a uniform row and a row at the fp16 ceiling."""
    g = _load(golden_dir, "rope_cache_softmax.npz")
    probs, tok = oracle.softmax_argmax(g["sm_logits"])
    ref = g["sm_probs"].astype(np.float64)
    assert np.array_equal(tok, g["sm_argmax"])
    assert np.abs(probs - ref).max() <= 1e-3
    big = ref > 1e-30
    rel = np.abs(probs[big] - ref[big]) / ref[big]
    assert rel.max() < 1e-5, rel.max()
    assert np.allclose(np.log(np.maximum(probs, 1e-45))[big], g["sm_logprobs"][big], atol=1e-4)


# ----------------------------------------------------------------- a3 / a6 pinned to the reference's own Python statements

def test_rowabsmax_quant_matches_reference_sym_quant_bit_exact(oracle, golden_dir):
    """a3: the oracle's restatement of rowAbsMaxQuantizeKernel (quant.cu:102-167) against the reference's Python statement of
    the same quantiser -- scales = (max|x| / 7).to(fp16) * clip (quarot/nn/quantization.py:10), sym_quant + pack_i4
    (quarot/functional/quantization.py:29-49) -- run on CPU by tests/golden/make_golden.py: packed bytes and scales bit for bit."""
    g = _load(golden_dir, "sym_quant_w4a16.npz")
    for idx in range(int(g["sq_cases"])):
        x, clip = g[f"sq{idx}_x"], float(g[f"sq{idx}_clip"])
        q, scale = oracle.rowabsmax_quant_i4(x, clip)
        assert np.array_equal(scale.view(np.uint16), g[f"sq{idx}_scale"].view(np.uint16)), idx
        assert np.array_equal(q.view(np.uint8), g[f"sq{idx}_q"].view(np.uint8)), idx


def test_w4a16_against_reference_dequantised_matmul(oracle, golden_dir):
    """a6: the oracle's W4A16 -- out = h((sum_k x w) * s), fp64 accumulate, ONE rounding -- against the reference's own
    statement of forward_w4a16: unpack_i4(weight).to(fp16) * scales (an fp16 weight: s * w ROUNDED to fp16, which is also what
    BitBLAS' with_scaling decode does), then the matmul (quarot_nn/linear.py:111-119; fp32 accumulation as
    quarot_nn/qspec_gemm.py:20-88), run on CPU by make_golden.py.
    What the pin shows (and the bars below hold it there): the two statements differ by the fp16 rounding of every s * w --
    2^-12 relative per weight, ~sqrt(K) 2^-12 of a term on the sum -- i.e. up to 2e-3 of max(1, |out|) at K = 14 k, 0.2-2.5 %
    of the elements above 1e-3.  Measured against the EXACT value of sum_k x (s w) (fp64, nothing rounded) the oracle's form is
    the closer one: its only error is the final fp16 rounding.  The product computes the oracle's form."""
    g = _load(golden_dir, "sym_quant_w4a16.npz")
    for idx in range(int(g["wa_cases"])):
        x, wq, ws = g[f"wa{idx}_x"], g[f"wa{idx}_wq"], g[f"wa{idx}_ws"]
        out = oracle.gemm_w4a16(x, wq, ws).astype(np.float64)
        w = oracle.unpack_i4(wq).astype(np.float64)
        exact = x.astype(np.float64) @ (w * ws.astype(np.float64)[:, None]).T
        den = np.maximum(1.0, np.abs(exact))
        e_or = np.abs(out - exact) / den
        assert e_or.max() <= 2.0 ** -11 * 1.01, (idx, e_or.max())                  # the one fp16 rounding, nothing else
        for name in ("ref_f32", "ref_f16"):
            ref = g[f"wa{idx}_{name}"].astype(np.float64)
            r = np.abs(out - ref) / np.maximum(1.0, np.abs(ref))
            assert (r <= 1e-3).mean() >= 0.97 and r.max() <= 2.5e-3, (idx, name, r.max(), (r > 1e-3).mean())
            e_ref = np.abs(ref - exact) / den
            if x.shape[1] >= 4096:   # the reference form's weight rounding shows from K ~ 4 k on
                assert np.sqrt((e_or ** 2).mean()) <= np.sqrt((e_ref ** 2).mean()) * 1.5, (idx, name)


def test_typical_acceptance_matches_reference_run(oracle, golden_dir):
    """The oracle's TypicalAcceptanceSampler against the reference class itself run on CPU (make_golden.py: deterministic, 20
    cases over four flavours): output layout, accept masks and the three counters, exactly."""
    g = _load(golden_dir, "typical_acceptance.npz")
    n = int(g["cases"])
    assert n == 20
    seen = set()
    for i in range(n):
        thr, alpha = g[f"t{i}_params"]
        out, acc, rec, c, H = oracle.typical_acceptance_sample(g[f"t{i}_tq"], g[f"t{i}_bonus"], g[f"t{i}_ids"], thr, alpha)
        assert np.array_equal(out, g[f"t{i}_out"]), i
        assert np.array_equal(acc, g[f"t{i}_accepted"]), i
        assert list(c) == list(g[f"t{i}_counters"]), i
        seen |= set(np.unique(acc).tolist())
    assert seen == {False, True}


def test_sampling_top_k_top_p_matches_reference_functions(oracle, golden_dir):
    """The oracle's non-greedy sampler against the reference's own _apply_top_k_top_p / _multinomial (compiled from the reference
    file and run on CPU by make_golden.py, exponential draws recorded): rows of distinct logits -- keep masks and tokens exact,
    probabilities to 1e-6 --; rows with many equal logits -- the same outside the group of equal logits at the top-p boundary,
    which the reference's unstable sort splits arbitrarily."""
    g = _load(golden_dir, "sampling.npz")
    for name in ("distinct", "ties"):
        lg = g[name + "_logits"]
        pr, tok, keep = oracle.sample_top_k_top_p(lg, g[name + "_temperature"], g[name + "_top_k"], g[name + "_top_p"], g[name + "_E"])
        rk, rp, rt = g[name + "_keep"], g[name + "_probs"], g[name + "_token"]
        assert rk.sum(1).min() >= 1 and (rk.sum(1) < lg.shape[1]).any()
        for t in range(lg.shape[0]):
            if name == "distinct":
                assert np.array_equal(keep[t], rk[t]), t
                assert np.abs(pr[t] - rp[t]).max() <= 1e-6 and tok[t] == rt[t], t
            else:
                b = lg[t][rk[t]].min()                       # the boundary group: the lowest logit the reference kept
                off = lg[t] != b
                assert np.array_equal(keep[t][off], rk[t][off]), t
