"""The fragment-major activation tiles of the verify pass (include/qspec_hip.h, "activation layout").

Every `_xp` entry must compute exactly the bits of its row-major twin (already checked against the oracle in
test_kernels_gpu.py): producers are compared through the documented offset formula, restated here in numpy;
consumers are fed a tile built with the same formula from the row-major input.
"""
import numpy as np
import pytest
import torch

from test_kernels_gpu import DEV, dev, host, make_paged, rand_hidden, rand_w4  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from qspec_amd import ops as o
    return o


def xp_offsets(K):
    """offset(r, k) of include/qspec_hip.h for the 16 x K tile -> int64 [16, K]."""
    k = np.arange(K)[None, :]
    r = np.arange(16)[:, None]
    kstep, rem = k // 128, k % 128
    g, dd, e = rem // 32, (rem // 8) % 4, k % 8
    pos = 2 * (e % 4) + e // 4
    return ((((kstep * 4 + dd) * 64) + g * 16 + r) * 8 + pos).astype(np.int64)


def test_offsets_are_a_permutation():
    for K in (128, 3584, 4096):
        off = xp_offsets(K)
        assert np.array_equal(np.sort(off.ravel()), np.arange(16 * K))


def to_tile(x):
    """Row-major [M <= 16, K] fp16 (numpy) -> device tile [16, K] (rows >= M carry NaN: they must never reach a result)."""
    M, K = x.shape
    flat = np.full(16 * K, np.nan, np.float16)
    flat[xp_offsets(K)[:M].ravel()] = x.ravel()
    return dev(flat.reshape(16, K))


def from_tile(t, M):
    """Device tile [16, K] -> row-major numpy [M, K]."""
    K = t.shape[1]
    return host(t).ravel()[xp_offsets(K)[:M]]


def same_bits(a, b):
    a = a if isinstance(a, np.ndarray) else host(a)
    b = b if isinstance(b, np.ndarray) else host(b)
    return np.array_equal(np.ascontiguousarray(a).view(np.uint16), np.ascontiguousarray(b).view(np.uint16))


def tile(K):
    return torch.full((16, K), float("nan"), dtype=torch.float16, device=DEV)


@pytest.mark.parametrize("M,K,ok", [(1, 4096, True), (16, 4096, True), (16, 5120, True), (8, 3584, True), (3, 4608, True),
                                    (17, 4096, False), (8, 8192, False), (8, 2048, False)])
def test_supported_shapes(ops, M, K, ok):
    assert ops.w4a16_act_layout_supported(M, K) == ok


# ------------------------------------------------------------------ producers

@pytest.mark.parametrize("T,H,with_delta", [(1, 4096, False), (8, 4096, True), (16, 4096, True), (5, 5120, True)])
def test_norm_writes_the_tile(ops, T, H, with_delta):
    rng = np.random.default_rng(T + H)
    x = dev(rand_hidden(rng, T, H))
    delta = dev(rand_hidden(rng, T, H)) if with_delta else None
    n0 = torch.empty(T, H, dtype=torch.float16, device=DEV); h0 = torch.empty_like(n0)
    ops.add_rms_norm_fp16(n0, h0, x, delta, 1e-5)
    n1 = tile(H); h1 = torch.empty_like(n0)
    ops.add_rms_norm_fp16(n1, h1, x, delta, 1e-5, xp=True)
    assert same_bits(from_tile(n1, T), n0)
    if with_delta:
        assert same_bits(h0, h1)


@pytest.mark.parametrize("T,H,S", [(8, 4096, 4), (16, 4096, 2), (3, 5120, 3)])
def test_norm_with_k_slice_finish_writes_the_tile(ops, T, H, S):
    rng = np.random.default_rng(T * 3 + H)
    x = dev(rand_hidden(rng, T, H))
    part = dev((rng.standard_normal((S, T, H)) * 300).astype(np.float32))
    ws = dev((rng.random(H) * 0.002 + 0.0005).astype(np.float16))
    n0 = torch.empty(T, H, dtype=torch.float16, device=DEV); h0 = torch.empty_like(n0)
    ops.add_rms_norm_fp16_partial(n0, h0, x, part, ws, S, 1e-5)
    n1 = tile(H); h1 = torch.empty_like(n0)
    ops.add_rms_norm_fp16_partial(n1, h1, x, part, ws, S, 1e-5, xp=True)
    assert same_bits(from_tile(n1, T), n0) and same_bits(h0, h1)


def _attention_partials(ops, rng, nq, nkv, ctx_lens, q_len, n_splits):
    d, bs = 128, 16
    n_seqs = len(ctx_lens)
    bt, kc, vc = make_paged(rng, n_seqs, ctx_lens, nkv, d, bs)
    T = n_seqs * q_len
    row = (nq + 2 * nkv) * d
    qkv = dev((rng.standard_normal((T, row)) * 0.5).astype(np.float16))
    q_start = dev((np.arange(n_seqs + 1) * q_len).astype(np.int32))
    ws = torch.zeros(ops.paged_attention_workspace_bytes(T, nq, d, n_splits), dtype=torch.uint8, device=DEV)
    ops.paged_attention(qkv, row, dev(kc), dev(vc), dev(bt), dev(np.array(ctx_lens, np.int32)), q_start, T, q_len, nq,
                        d ** -0.5, n_splits, ws, None)
    return ws, T


@pytest.mark.parametrize("ctx_lens,q_len,n_splits", [([37, 128, 129, 500], 4, 8), ([600], 1, 8), ([77, 300], 4, 5)])
def test_head_transform_writes_the_tile(ops, oracle, ctx_lens, q_len, n_splits):
    rng = np.random.default_rng(sum(ctx_lens) + q_len)
    nq, d = 32, 128
    ws, T = _attention_partials(ops, rng, nq, 8, ctx_lens, q_len, n_splits)
    sc = oracle.rsqrt_scale(nq)
    o0 = torch.empty(T, nq * d, dtype=torch.float16, device=DEV)
    ops.heads_hadamard_merged(ws, T, n_splits, T, nq, d, sc, out_f16=o0)
    o1 = tile(nq * d)
    ops.heads_hadamard_merged(ws, T, n_splits, T, nq, d, sc, out_f16=o1, xp=True)
    assert same_bits(from_tile(o1, T), o0)


def test_table_factor_head_transform_writes_the_tile(ops, oracle):
    from qspec_amd import hadamard_tables
    from test_kernels_gpu import _attn_partials_ws
    rng = np.random.default_rng(40)
    T, nq, d = 12, 40, 128
    hadK, K = hadamard_tables.get_hadK(nq)
    hk = hadK.to(torch.float16).to(DEV)
    assert ops.heads_hadamard_mix_merged_spread_supported(T, nq, d, K)
    o = rng.standard_normal((T, nq, 2, d)) * 2.0
    m = rng.standard_normal((T, nq, 2)) * 3.0
    l = rng.random((T, nq, 2)) * 5.0 + 0.5
    ws = _attn_partials_ws(ops, T, nq, d, 2, o, m, l)
    sc = oracle.rsqrt_scale(nq)
    o0 = torch.empty(T, nq * d, dtype=torch.float16, device=DEV)
    ops.heads_hadamard_mix_merged_spread(ws, T, 2, T, nq, d, hk, K, sc, o0)
    o1 = tile(nq * d)
    ops.heads_hadamard_mix_merged_spread(ws, T, 2, T, nq, d, hk, K, sc, o1, xp=True)
    assert same_bits(from_tile(o1, T), o0)


@pytest.mark.parametrize("T,I,K", [(8, 14336, 28), (16, 14336, 28), (3, 13824, 108)])
def test_mlp_transform_writes_the_tile(ops, oracle, T, I, K):
    from qspec_amd import hadamard_tables
    rng = np.random.default_rng(T + I)
    hadK, K2 = hadamard_tables.get_hadK(I)
    assert K2 == K
    had = hadK.to(torch.float16).to(DEV)
    act = dev(rand_hidden(rng, T, I, 0.5))
    assert ops.mlp_hadamard_act_layout_supported(T, I, K)
    sc = oracle.rsqrt_scale(I)
    o0 = torch.empty(T, I, dtype=torch.float16, device=DEV)
    ops.mlp_hadamard(act, had, K, sc, out_f16=o0)
    o1 = tile(I)
    ops.mlp_hadamard(act, had, K, sc, out_f16=o1, xp=True)
    assert same_bits(from_tile(o1, T), o0)


# ------------------------------------------------------------------ consumers

@pytest.mark.parametrize("M,N,K", [(16, 4096, 4096), (5, 4096, 4096), (1, 6144, 4096), (12, 5120, 5120), (16, 1024, 3584),
                                   (7, 512, 4608)])
def test_linear_reads_the_tile(ops, oracle, M, N, K):
    rng = np.random.default_rng(M + N + K)
    x = rand_hidden(rng, M, K)
    wq = dev(oracle.pack_i4(rand_w4(rng, N, K)))
    ws = dev((rng.random(N) * 0.002 + 0.0005).astype(np.float16))
    o0 = torch.empty(M, N, dtype=torch.float16, device=DEV); o1 = torch.empty_like(o0)
    ops.w4a16_linear(dev(x), wq, ws, o0)
    ops.w4a16_linear(to_tile(x), wq, ws, o1, xp=True, tokens=M)
    assert same_bits(o0, o1)


@pytest.mark.parametrize("M,N,K", [(16, 4096, 14336), (5, 4096, 14336), (9, 5120, 13824)])
def test_k_sliced_linear_reads_the_tile(ops, oracle, M, N, K):
    rng = np.random.default_rng(M + N)
    x = rand_hidden(rng, M, K)
    wq = dev(oracle.pack_i4(rand_w4(rng, N, K)))
    S = ops.w4a16_linear_partial_slices(M, N, K)
    p0 = torch.empty(S, M, N, dtype=torch.float32, device=DEV); p1 = torch.empty_like(p0)
    ops.w4a16_linear_partial(dev(x), wq, p0, S)
    ops.w4a16_linear_partial(to_tile(x), wq, p1, S, xp=True, tokens=M)
    torch.cuda.synchronize()
    assert torch.equal(p0.view(torch.int32), p1.view(torch.int32))


@pytest.mark.parametrize("M", [1, 8, 16])
def test_fused_qkv_and_gate_up_read_the_tile(ops, oracle, M):
    rng = np.random.default_rng(M)
    K, nq, nkv, d, bs, I = 4096, 32, 8, 128, 16, 14336
    N = (nq + 2 * nkv) * d
    x = rand_hidden(rng, M, K)
    wq = dev(oracle.pack_i4(rand_w4(rng, N, K)))
    ws = dev((rng.random(N) * 0.002 + 0.0005).astype(np.float16))
    pos = dev(rng.integers(0, 2000, M).astype(np.int64))
    csc = dev((rng.standard_normal((2048, d))).astype(np.float16))
    slots = dev(rng.permutation(64 * bs)[:M].astype(np.int64))
    outs = []
    for xp in (False, True):
        kc = torch.zeros(64, bs, nkv, d, dtype=torch.float16, device=DEV); vc = torch.zeros_like(kc)
        qkv = torch.empty(M, N, dtype=torch.float16, device=DEV)
        ops.qkv_rope_linear(to_tile(x) if xp else dev(x), None, wq, ws, qkv, pos, csc, kc, vc, slots, nq, nkv, d, xp=xp, tokens=M)
        outs.append((qkv, kc, vc))
    for a, b in zip(*outs):
        assert same_bits(a, b)
    gw = dev(oracle.pack_i4(rand_w4(rng, 2 * I, K)))
    gs = dev((rng.random(2 * I) * 0.002 + 0.0005).astype(np.float16))
    a0 = torch.empty(M, I, dtype=torch.float16, device=DEV); a1 = torch.empty_like(a0)
    ops.gate_up_silu_linear(dev(x), None, gw, gs, a0)
    ops.gate_up_silu_linear(to_tile(x), None, gw, gs, a1, xp=True, tokens=M)
    assert same_bits(a0, a1)


def test_unsupported_shapes_fail_loudly(ops, oracle):
    rng = np.random.default_rng(3)
    x = to_tile(rand_hidden(rng, 4, 8192))
    wq = dev(oracle.pack_i4(rand_w4(rng, 256, 8192)))
    ws = dev(np.ones(256, np.float16))
    with pytest.raises(RuntimeError):
        ops.w4a16_linear(x, wq, ws, torch.empty(4, 256, dtype=torch.float16, device=DEV), xp=True, tokens=4)
    with pytest.raises(RuntimeError):   # a tile always has 16 rows
        ops.add_rms_norm_fp16(torch.empty(4, 4096, dtype=torch.float16, device=DEV), None, dev(rand_hidden(rng, 4, 4096)), None,
                              1e-5, xp=True)


# ------------------------------------------------------------------ 17..32 tokens: two tiles, the two-token-tile streaming kernel

def to_tiles32(x):
    """Row-major [M <= 32, K] fp16 (numpy) -> device [2, 16, K]: rows 0..15 and 16..31 as fragment-major tiles (NaN padding)."""
    M, K = x.shape
    flat = np.full(2 * 16 * K, np.nan, np.float16)
    off = xp_offsets(K)
    for t in range(2):
        rows = x[16 * t: min(M, 16 * t + 16)]
        if len(rows):
            flat[16 * K * t + off[:len(rows)].ravel()] = rows.ravel()
    return dev(flat.reshape(2, 16, K))


@pytest.mark.parametrize("M,N,K", [(32, 4096, 4096), (17, 1024, 4096), (24, 10240, 8192), (32, 8192, 8192), (32, 2048, 28672),
                                   (20, 4096, 14336), (32, 5120, 5120), (32, 57344, 8192)])
def test_two_tile_linear_within_1e3(ops, oracle, M, N, K):
    """The two-token-tile W4A16 streaming kernel (K in 1 / 2 / 7 passes) against the oracle, and deterministic."""
    assert ops.w4a16_act_layout32_supported(M, N, K)
    rng = np.random.default_rng(M + N + K)
    x = rand_hidden(rng, M, K)
    wq_np = oracle.pack_i4(rand_w4(rng, N, K))
    ws_np = (rng.random(N) * 0.002 + 0.0005).astype(np.float16)
    wq, ws = dev(wq_np), dev(ws_np)
    out = torch.full((M + 1, N), 7.0, dtype=torch.float16, device=DEV)
    ops.w4a16_linear_xp32(to_tiles32(x), wq, ws, out[:M], M)
    if N * K <= 10240 * 8192:
        from test_kernels_gpu import assert_close_1e3
        assert_close_1e3(host(out[:M]), oracle.gemm_w4a16(x, wq_np, ws_np))
    ref = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.w4a16_linear(dev(x), wq, ws, ref)     # the M-tiled kernel: another summation order
    d = (out[:M].float() - ref.float()).abs() / (1e-3 * ref.float().abs().clamp(min=1.0))
    assert float(d.max()) <= 2.0, float(d.max())
    assert torch.all(out[M] == 7.0)
    again = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.w4a16_linear_xp32(to_tiles32(x), wq, ws, again, M)
    assert same_bits(again, out[:M])


@pytest.mark.parametrize("M,K,nq,nkv,I", [(32, 4096, 32, 8, 14336), (19, 8192, 16, 2, 3584)])
def test_two_tile_fused_epilogues(ops, oracle, M, K, nq, nkv, I):
    """qkv + RoPE + KV write and gate_up + silu on the two-tile kernel == its plain form followed by the separate ops."""
    rng = np.random.default_rng(M + K)
    d, bs = 128, 16
    N = (nq + 2 * nkv) * d
    x = to_tiles32(rand_hidden(rng, M, K))
    wq = dev(oracle.pack_i4(rand_w4(rng, N, K)))
    ws = dev((rng.random(N) * 0.002 + 0.0005).astype(np.float16))
    pos = dev(rng.integers(0, 2000, M).astype(np.int64))
    cs = dev(oracle.make_cos_sin_cache(d, 2048, 10000.0))
    slots_np = rng.permutation(64 * bs)[:M].astype(np.int64)
    slots_np[1] = -1
    slots = dev(slots_np)
    ref = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.w4a16_linear_xp32(x, wq, ws, ref, M)
    kc0 = torch.zeros(64, bs, nkv, d, dtype=torch.float16, device=DEV); vc0 = torch.zeros_like(kc0)
    ops.rope_kv_write(pos, ref, cs, kc0, vc0, slots, nq, nkv, d)
    out = torch.empty_like(ref)
    kc1 = torch.zeros_like(kc0); vc1 = torch.zeros_like(kc0)
    ops.qkv_rope_linear_xp32(x, wq, ws, out, pos, cs, kc1, vc1, slots, nq, nkv, d, M)
    torch.cuda.synchronize()
    assert same_bits(out, ref) and torch.equal(kc1, kc0) and torch.equal(vc1, vc0)
    gw = dev(oracle.pack_i4(rand_w4(rng, 2 * I, K)))
    gs = dev((rng.random(2 * I) * 0.002 + 0.0005).astype(np.float16))
    gu = torch.empty(M, 2 * I, dtype=torch.float16, device=DEV)
    ops.w4a16_linear_xp32(x, gw, gs, gu, M)
    a0 = ops.silu_mul(gu, torch.empty(M, I, dtype=torch.float16, device=DEV))
    a1 = ops.gate_up_silu_linear_xp32(x, gw, gs, torch.empty(M, I, dtype=torch.float16, device=DEV), M)
    assert same_bits(a0, a1)


def from_tiles32(t, M):
    """Device [2, 16, K] (two fragment-major tiles) -> row-major numpy [M, K]."""
    K = t.shape[-1]
    flat = host(t).ravel()
    off = xp_offsets(K)
    rows = [flat[16 * K * (r >> 4) + off[r & 15]] for r in range(M)]
    return np.stack(rows)


@pytest.mark.parametrize("T,H", [(17, 4096), (32, 8192), (24, 5120)])
def test_norm_writes_two_tiles(ops, T, H):
    rng = np.random.default_rng(T + H)
    x = dev(rand_hidden(rng, T, H)); delta = dev(rand_hidden(rng, T, H))
    n0 = torch.empty(T, H, dtype=torch.float16, device=DEV); h0 = torch.empty_like(n0)
    ops.add_rms_norm_fp16(n0, h0, x, delta, 1e-5)
    n1 = torch.full((2, 16, H), float("nan"), dtype=torch.float16, device=DEV); h1 = torch.empty_like(n0)
    ops.add_rms_norm_fp16(n1, h1, x, delta, 1e-5, xp=True)
    assert same_bits(from_tiles32(n1, T), n0) and same_bits(h0, h1)
    S = 3
    part = dev((rng.standard_normal((S, T, H)) * 300).astype(np.float32))
    ws = dev((rng.random(H) * 0.002 + 0.0005).astype(np.float16))
    ops.add_rms_norm_fp16_partial(n0, h0, x, part, ws, S, 1e-5)
    n1.fill_(float("nan"))
    ops.add_rms_norm_fp16_partial(n1, h1, x, part, ws, S, 1e-5, xp=True)
    assert same_bits(from_tiles32(n1, T), n0) and same_bits(h0, h1)


@pytest.mark.parametrize("nq,nkv,ctx_lens,q_len", [(32, 8, [37, 128, 129, 500, 77], 4), (64, 8, [600, 40, 300, 90, 17, 260, 33, 511], 4),
                                                   (64, 8, [70, 200, 45], 4)])
def test_head_transform_writes_two_tiles(ops, oracle, nq, nkv, ctx_lens, q_len):
    """32 heads: the spread kernel; 64 heads (Llama-3-70B): the one-workgroup kernel's 8-byte groups."""
    rng = np.random.default_rng(sum(ctx_lens) + nq)
    d, n_splits = 128, 4
    ws, T = _attention_partials(ops, rng, nq, nkv, ctx_lens, q_len, n_splits)
    sc = oracle.rsqrt_scale(nq)
    o0 = torch.empty(T, nq * d, dtype=torch.float16, device=DEV)
    ops.heads_hadamard_merged(ws, T, n_splits, T, nq, d, sc, out_f16=o0)
    o1 = torch.full((2, 16, nq * d), float("nan"), dtype=torch.float16, device=DEV)
    ops.heads_hadamard_merged(ws, T, n_splits, T, nq, d, sc, out_f16=o1, xp=True)
    assert same_bits(from_tiles32(o1, T), o0)


@pytest.mark.parametrize("M,N,K", [(32, 8192, 28672), (20, 4096, 14336), (32, 1024, 8192)])
def test_two_tile_k_sliced_linear_and_norm_finish(ops, oracle, M, N, K):
    """down_proj at 17..32 tokens: one-pass K slices of the two-tile kernel across the workgroups, raw fp32 sums finished in the
    norm: hidden / normed rows within the W4A16 bar of w4a16_linear + add_rms_norm_fp16 (another summation order)."""
    rng = np.random.default_rng(M + N)
    x = rand_hidden(rng, M, K)
    wq = dev(oracle.pack_i4(rand_w4(rng, N, K)))
    ws = dev((rng.random(N) * 0.002 + 0.0005).astype(np.float16))
    S = ops.w4a16_linear_partial_slices_xp32(M, N, K)
    assert S == K // 4096 if K % 4096 == 0 and K // 4096 <= 8 else S == K // 2048
    part = torch.empty(S, M, N, dtype=torch.float32, device=DEV)
    ops.w4a16_linear_partial_xp32(to_tiles32(x), wq, part, S, M)
    hidden = dev(rand_hidden(rng, M, N))
    n1 = torch.empty_like(hidden); h1 = torch.empty_like(hidden)
    ops.add_rms_norm_fp16_partial(n1, h1, hidden, part, ws, S, 1e-5)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.w4a16_linear(dev(x), wq, ws, out)
    n0 = torch.empty_like(hidden); h0 = torch.empty_like(hidden)
    ops.add_rms_norm_fp16(n0, h0, hidden, out, 1e-5)
    torch.cuda.synchronize()
    d = (h1.float() - h0.float()).abs() / (1e-3 * h0.float().abs().clamp(min=1.0))
    assert float(d.max()) <= 2.0, float(d.max())
    assert (n1.float() - n0.float()).abs().max().item() < 2e-2


@pytest.mark.parametrize("T,I,K", [(32, 14336, 28), (20, 28672, 28), (17, 13824, 108)])
def test_mlp_transform_writes_two_tiles(ops, oracle, T, I, K):
    from qspec_amd import hadamard_tables
    rng = np.random.default_rng(T + I)
    hadK, K2 = hadamard_tables.get_hadK(I)
    assert K2 == K and ops.mlp_hadamard_act_layout_supported(T, I, K)
    had = hadK.to(torch.float16).to(DEV)
    act = dev(rand_hidden(rng, T, I, 0.5))
    sc = oracle.rsqrt_scale(I)
    o0 = torch.empty(T, I, dtype=torch.float16, device=DEV)
    ops.mlp_hadamard(act, had, K, sc, out_f16=o0)
    o1 = torch.full((2, 16, I), float("nan"), dtype=torch.float16, device=DEV)
    ops.mlp_hadamard(act, had, K, sc, out_f16=o1, xp=True)
    assert same_bits(from_tiles32(o1, T), o0)
