"""Generate the committed golden fixtures by RUNNING the importable pieces of the
reference (`/root/reference`, present only in the build container) on CPU.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz

What is imported from the reference (by file path; nothing is copied):
  * third-party/QuaRot/quarot/functional/quantization.py   pack_i4 / unpack_i4
  * third-party/QuaRot/quarot/functional/hadamard.py        get_hadK tables, matmul_hadU
      (needs the names `fast_hadamard_transform` / `fast_hadamard_transform_cuda` to exist at
       import time; they are CUDA extension modules absent here, so two EMPTY module objects are
       registered -- none of the functions called below touch them)
  * vllm/model_executor/layers/{spec_decode_base_sampler,rejection_sampler,typical_acceptance_sampler}.py
      RejectionSampler and TypicalAcceptanceSampler on CPU
      (imported under an empty package shell `vllm` providing only envs.VLLM_USE_FLASHINFER_SAMPLER,
       logger.init_logger and platforms.current_platform.simple_compile_backend = "eager")
  * third-party/ao/test/test_rowwise_scaled_linear_cutlass.py:64-84 is a formula, restated below
    with torch CPU ops on inputs of the shapes that test lists (:14-21).
  * tests/kernels/test_flash_attn.py:19-75   ref_paged_attn (paged KV, GQA, causal varlen) -- the module is loaded
      with `vllm.vllm_flash_attn` registered as an empty shell (a CUDA wheel absent here; ref_paged_attn is pure torch)
  * vllm/model_executor/layers/rotary_embedding.py:47-70,136-150,201-229   RotaryEmbedding._compute_cos_sin_cache,
      forward_native, _apply_rotary_emb -- loaded with a three-line `CustomOp` base (an nn.Module with a `register`
      decorator) and an empty `vllm._custom_ops` (forward_native touches neither)
  * vllm/model_executor/layers/sampler.py   _apply_top_k_top_p (:387-413) and _multinomial (:585-604): the module itself needs
      msgspec (absent); the two function definitions are compiled from the reference file and run (gen_sampling)
  * third-party/QuaRot/quarot/functional/quantization.py   sym_quant / sym_dequant (the a3 / a6 pins, gen_sym_quant_w4a16)
  * tests/kernels/test_cache.py:295-304 (reshape_and_cache_flash's reference loop) and
    vllm/model_executor/layers/sampler.py:278-287 (greedy probs / logprobs / argmax) are a five-line loop and three
    torch calls inside larger functions: restated below with the same torch CPU ops.

The fixtures are data (inputs + expected outputs); the generated .npz files are
small and committed so the tests run without the reference (it does not exist
on the GPU box).
"""
import importlib.util
import logging
import os
import sys
import types

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")  # the reference wraps _multinomial in torch.compile; run it eagerly

import numpy as np  # noqa: E402
import torch  # noqa: E402

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def gen_pack():
    qz = load_by_path("ref_quantization", f"{REF}/third-party/QuaRot/quarot/functional/quantization.py")
    g = torch.Generator().manual_seed(0)
    q = torch.randint(-8, 8, (7, 64), generator=g, dtype=torch.int8)
    q[0, :16] = torch.arange(-8, 8, dtype=torch.int8)
    packed = qz.pack_i4(q)
    unpacked = qz.unpack_i4(packed)
    assert torch.equal(unpacked.to(torch.int8), q)
    np.savez(os.path.join(OUT, "pack_i4.npz"), q=q.numpy(), packed=packed.numpy(), unpacked=unpacked.numpy())


def gen_hadamard():
    sys.modules.setdefault("fast_hadamard_transform", types.ModuleType("fast_hadamard_transform"))
    sys.modules.setdefault("fast_hadamard_transform_cuda", types.ModuleType("fast_hadamard_transform_cuda"))
    hd = load_by_path("ref_hadamard", f"{REF}/third-party/QuaRot/quarot/functional/hadamard.py")
    out = {}
    for K in (12, 20, 28, 36, 40, 52, 60, 108):
        hadK, k = hd.get_hadK(K)
        assert k == K
        out[f"had{K}"] = hadK.numpy().astype(np.int8)
    g = torch.Generator().manual_seed(1)
    for n in (32, 512, 14336, 28672, 13824, 40, 80):   # 40 / 80: head counts whose get_hadK factor is had40 (n/K = 1, 2)
        # fp16-representable inputs, so the fp64 result is the exact transform of what the fp16 kernels see
        x = torch.randn(3 if n <= 512 else 1, n, generator=g).to(torch.float16).to(torch.float64)
        y = hd.matmul_hadU(x)
        out[f"x_{n}"] = x.numpy()
        out[f"y_{n}"] = y.numpy()
    np.savez_compressed(os.path.join(OUT, "hadamard.npz"), **out)


def gen_w4a4():
    """Formula of third-party/ao/test/test_rowwise_scaled_linear_cutlass.py:64-84 on its own shapes."""
    out = {}
    g = torch.Generator().manual_seed(2)
    # M and K as in the reference test's list (:14-21); N trimmed to keep the committed fixture small
    cases = [(2, 512, 128), (3, 256, 2048), (4, 448, 640), (13, 64, 8576), (26, 128, 1664), (67, 64, 1408)]
    for idx, (m, n, k) in enumerate(cases):
        xq_s8 = torch.randint(-8, 8, (m, k), generator=g, dtype=torch.int8)
        wq_s8 = torch.randint(-8, 8, (n, k), generator=g, dtype=torch.int8)
        xs = (torch.rand(m, generator=g) * 0.1 + 0.01).to(torch.float16)
        ws = (torch.rand(n, generator=g) * 0.01 + 0.001).to(torch.float16)
        for use_bias in (False, True):
            bias = torch.rand(n, generator=g).to(torch.float16) if use_bias else None
            ref = (xq_s8.float() @ wq_s8.float().T) * xs.float().view(m, 1) * ws.float().view(1, n)
            if bias is not None:
                ref += bias.float()
            ref = ref.to(torch.float16)
            xq = (xq_s8[..., 1::2] << 4) | (xq_s8[..., 0::2] & 0xF)
            wq = (wq_s8[:, 1::2] << 4) | (wq_s8[:, 0::2] & 0xF)
            out[f"c{idx}_xq"] = xq.numpy()
            out[f"c{idx}_wq"] = wq.numpy()
            out[f"c{idx}_xs"] = xs.numpy()
            out[f"c{idx}_ws"] = ws.numpy()
            if bias is not None:
                out[f"c{idx}_bias"] = bias.numpy()
            out[f"c{idx}_ref_{'b' if use_bias else 'n'}"] = ref.numpy()
    np.savez_compressed(os.path.join(OUT, "w4a4_gemm.npz"), **out)


def import_ref_sampler():
    vllm = types.ModuleType("vllm")
    vllm.__path__ = []
    envs = types.ModuleType("vllm.envs")
    envs.VLLM_USE_FLASHINFER_SAMPLER = False
    logger = types.ModuleType("vllm.logger")
    logger.init_logger = lambda name: logging.getLogger(name)
    platforms = types.ModuleType("vllm.platforms")
    platforms.current_platform = types.SimpleNamespace(simple_compile_backend="eager")
    me = types.ModuleType("vllm.model_executor")
    me.__path__ = []
    layers = types.ModuleType("vllm.model_executor.layers")
    layers.__path__ = []
    for name, mod in (("vllm", vllm), ("vllm.envs", envs), ("vllm.logger", logger), ("vllm.platforms", platforms),
                      ("vllm.model_executor", me), ("vllm.model_executor.layers", layers)):
        sys.modules[name] = mod
    vllm.envs = envs
    load_by_path("vllm.model_executor.layers.spec_decode_base_sampler",
                 f"{REF}/vllm/model_executor/layers/spec_decode_base_sampler.py")
    rs = load_by_path("vllm.model_executor.layers.rejection_sampler",
                      f"{REF}/vllm/model_executor/layers/rejection_sampler.py")
    return rs


def gen_rejection():
    """Run the reference RejectionSampler on CPU with the random draws replaced by recorded arrays."""
    rs = import_ref_sampler()
    out = {}
    g = torch.Generator().manual_seed(3)
    cases = [(4, 3, 257), (3, 5, 1000), (10, 5, 512), (1, 1, 64), (8, 4, 1024)]
    for idx, (B, k, V) in enumerate(cases):
        for flavour in ("random", "agree", "onehot"):
            if flavour == "random":
                tq = torch.softmax(torch.randn(B, k + 1, V, generator=g) * 2, -1)
                dp = torch.softmax(torch.randn(B, k, V, generator=g) * 2, -1)
            elif flavour == "agree":  # draft close to target: mostly accepted
                logits = torch.randn(B, k + 1, V, generator=g) * 3
                tq = torch.softmax(logits, -1)
                dp = torch.softmax(logits[:, :k] + 0.3 * torch.randn(B, k, V, generator=g), -1)
            else:  # one-hot draft, one-hot target (reference's own test style, test_rejection_sampler.py:48-127)
                tq = torch.zeros(B, k + 1, V)
                tq[torch.arange(B)[:, None], torch.arange(k + 1)[None], torch.randint(0, V, (B, k + 1), generator=g)] = 1
                dp = torch.zeros(B, k, V)
                dp[torch.arange(B)[:, None], torch.arange(k)[None], torch.randint(0, V, (B, k), generator=g)] = 1
            ids = torch.multinomial(dp.reshape(-1, V), 1, generator=g).reshape(B, k)
            if flavour == "onehot":  # make about half of the proposals agree with the target
                agree = torch.rand(B, k, generator=g) < 0.5
                ids = torch.where(agree, tq[:, :k].argmax(-1), ids)
                dp = torch.zeros(B, k, V)
                dp[torch.arange(B)[:, None], torch.arange(k)[None], ids] = 1
            bonus = torch.randint(0, V, (B, 1), generator=g)
            U = torch.rand(B, k, generator=g)
            E = torch.empty(B * k, V).exponential_(1.0, generator=g)
            sampler = rs.RejectionSampler()
            sampler.init_tensors(device="cpu", device_type="cpu")
            orig_rand, orig_exp = torch.rand, torch.Tensor.exponential_
            try:
                torch.rand = lambda *a, **kw: U.clone()
                torch.Tensor.exponential_ = lambda self, *a, **kw: self.copy_(E)
                o = sampler(tq, bonus, dp, ids)
            finally:
                torch.rand, torch.Tensor.exponential_ = orig_rand, orig_exp
            key = f"c{idx}_{flavour}"
            out[key + "_tq"] = tq.numpy().astype(np.float32)
            out[key + "_dp"] = dp.numpy().astype(np.float32)
            out[key + "_ids"] = ids.numpy()
            out[key + "_bonus"] = bonus.numpy()
            out[key + "_U"] = U.numpy()
            out[key + "_E"] = E.reshape(B, k, V).numpy()
            out[key + "_out"] = o.numpy()
            out[key + "_counters"] = np.array([int(sampler.num_accepted_tokens), int(sampler.num_emitted_tokens),
                                               int(sampler.num_draft_tokens)], np.int64)
    # _create_output known-answer cases in the style of tests/samplers/test_rejection_sampler.py:48-127
    sampler = rs.RejectionSampler()
    sampler.init_tensors(device="cpu", device_type="cpu")
    B, k, V = 10, 5, 3000
    for name, last in (("all", torch.full((B,), k - 1)), ("none", torch.full((B,), -1)),
                       ("some", torch.randint(-1, k, (B,), generator=g))):
        accepted = torch.arange(k)[None, :] <= last[:, None]
        if name == "some":  # non-causal extra True values after the first False must be ignored
            accepted[0, -1] = True
        rec = torch.randint(0, V, (B, k), generator=g)
        ids = torch.randint(0, V, (B, k), generator=g)
        bonus = torch.randint(0, V, (B, 1), generator=g)
        o = sampler._create_output(accepted, rec, ids, bonus)
        out[f"co_{name}_accepted"] = accepted.numpy()
        out[f"co_{name}_rec"] = rec.numpy()
        out[f"co_{name}_ids"] = ids.numpy()
        out[f"co_{name}_bonus"] = bonus.numpy()
        out[f"co_{name}_out"] = o.numpy()
    np.savez_compressed(os.path.join(OUT, "rejection.npz"), **out)


def _vllm_shell():
    """Empty package shell `vllm` with the few attributes the loaded reference files read at import time."""
    if "vllm" in sys.modules and hasattr(sys.modules["vllm"], "_qspec_shell"):
        return sys.modules["vllm"]
    vllm = types.ModuleType("vllm")
    vllm.__path__ = []
    vllm._qspec_shell = True
    platforms = types.ModuleType("vllm.platforms")
    platforms.current_platform = types.SimpleNamespace(simple_compile_backend="eager",
                                                       seed_everything=lambda seed: torch.manual_seed(seed))
    fa = types.ModuleType("vllm.vllm_flash_attn")           # CUDA wheel, absent: names only
    fa.flash_attn_varlen_func = fa.flash_attn_with_kvcache = None
    cops = types.ModuleType("vllm._custom_ops")             # CUDA extension bindings, absent
    me = types.ModuleType("vllm.model_executor")
    me.__path__ = []
    co = types.ModuleType("vllm.model_executor.custom_op")

    class CustomOp(torch.nn.Module):
        @classmethod
        def register(cls, name):
            return lambda op_cls: op_cls
    co.CustomOp = CustomOp
    layers = types.ModuleType("vllm.model_executor.layers")
    layers.__path__ = []
    for name, mod in (("vllm", vllm), ("vllm.platforms", platforms), ("vllm.vllm_flash_attn", fa),
                      ("vllm._custom_ops", cops), ("vllm.model_executor", me), ("vllm.model_executor.custom_op", co),
                      ("vllm.model_executor.layers", layers)):
        sys.modules[name] = mod
    vllm._custom_ops = cops
    vllm.platforms = platforms
    return vllm


def gen_attention():
    """ref_paged_attn of the reference's own flash-attn test, run on CPU.

    Two outputs per case: `ref16` = the function on fp16 tensors exactly as the reference test calls it (its test bar
    against flash-attn is atol 2e-2 / rtol 1e-2, test_flash_attn.py:155-156,  because S and P are rounded to fp16 in
    it), and `ref32` = the same function on the same fp16-representable values held as fp32 tensors, which makes it
    the exact statement of the paged / GQA / causal-varlen semantics the 1e-3 comparisons need."""
    _vllm_shell()
    fa = load_by_path("ref_test_flash_attn", f"{REF}/tests/kernels/test_flash_attn.py")
    out = {}
    g = torch.Generator().manual_seed(4)
    # (query heads, kv heads, head size, block size, num blocks, kv_lens, query_lens)
    cases = [(8, 2, 128, 16, 24, [70, 18, 133], [4, 1, 6]),       # verify-shaped (k+1 queries) and decode rows, GQA 4
             (4, 1, 64, 16, 12, [1, 54, 100], [1, 4, 4]),         # head_dim 64 (TinyLlama), first-token row
             (8, 2, 128, 32, 8, [128, 97], [1, 1])]               # decode only, block size 32
    for idx, (nq, nkv, d, bs, nb, kv_lens, q_lens) in enumerate(cases):
        T = sum(q_lens)
        q = torch.randn(T, nq, d, generator=g).to(torch.float16)
        kc = torch.randn(nb, bs, nkv, d, generator=g).to(torch.float16)
        vc = torch.randn(nb, bs, nkv, d, generator=g).to(torch.float16)
        max_blocks = (max(kv_lens) + bs - 1) // bs
        bt = torch.stack([torch.randperm(nb, generator=g)[:max_blocks] for _ in kv_lens]).to(torch.int32)
        scale = d ** -0.5
        ref16 = fa.ref_paged_attn(q.clone(), kc, vc, q_lens, kv_lens, bt, scale)      # q is scaled in place: clone
        ref32 = fa.ref_paged_attn(q.float(), kc.float(), vc.float(), q_lens, kv_lens, bt, scale)
        assert ref16.dtype == torch.float16 and ref32.dtype == torch.float32
        key = f"c{idx}_"
        out[key + "q"] = q.numpy()
        out[key + "key_cache"] = kc.numpy()
        out[key + "value_cache"] = vc.numpy()
        out[key + "block_tables"] = bt.numpy()
        out[key + "kv_lens"] = np.array(kv_lens, np.int32)
        out[key + "query_lens"] = np.array(q_lens, np.int32)
        out[key + "scale"] = np.array(scale, np.float32)
        out[key + "ref16"] = ref16.numpy()
        out[key + "ref32"] = ref32.numpy()
    np.savez_compressed(os.path.join(OUT, "attention.npz"), **out)


def gen_rope_cache_softmax():
    _vllm_shell()
    re_ = load_by_path("vllm.model_executor.layers.rotary_embedding",
                       f"{REF}/vllm/model_executor/layers/rotary_embedding.py")
    out = {}
    g = torch.Generator().manual_seed(5)
    # ---- RotaryEmbedding._compute_cos_sin_cache + forward_native (neox style, fp16), as quarot_llama.py:106-114 builds it
    for idx, (d, max_pos, base, nq, nkv, T) in enumerate([(128, 2048, 500000.0, 8, 2, 7), (64, 512, 10000.0, 4, 4, 5)]):
        rot = re_.RotaryEmbedding(d, d, max_pos, base, True, torch.float16)
        pos = torch.randint(0, max_pos, (T,), generator=g)
        pos[0], pos[-1] = 0, max_pos - 1
        q = torch.randn(T, nq * d, generator=g).to(torch.float16)
        k = torch.randn(T, nkv * d, generator=g).to(torch.float16)
        qo, ko = rot.forward_native(pos, q.clone(), k.clone())
        key = f"rope{idx}_"
        out[key + "cfg"] = np.array([d, max_pos, nq, nkv], np.int64)
        out[key + "base"] = np.array(base, np.float64)
        # the whole table is a function of (d, max_pos, base): keep a strided sample of rows + the rows `pos` uses
        rows = torch.unique(torch.cat([torch.arange(0, max_pos, 37), pos]))
        out[key + "cache_rows"] = rows.numpy()
        out[key + "cache"] = rot.cos_sin_cache[rows].numpy()
        out[key + "pos"] = pos.numpy()
        out[key + "q"], out[key + "k"] = q.numpy(), k.numpy()
        out[key + "q_out"], out[key + "k_out"] = qo.numpy(), ko.numpy()
    # ---- reshape_and_cache_flash reference loop (tests/kernels/test_cache.py:295-304)
    T, nkv, d, bs, nb = 9, 2, 128, 16, 6
    slot_mapping = torch.randperm(bs * nb, generator=g)[:T]
    key_t = torch.randn(T, nkv, d, generator=g).to(torch.float16)
    val_t = torch.randn(T, nkv, d, generator=g).to(torch.float16)
    kc = torch.randn(nb, bs, nkv, d, generator=g).to(torch.float16)
    vc = torch.randn(nb, bs, nkv, d, generator=g).to(torch.float16)
    out["cache_key"], out["cache_value"] = key_t.numpy(), val_t.numpy()
    out["cache_key_cache_in"], out["cache_value_cache_in"] = kc.numpy().copy(), vc.numpy().copy()
    out["cache_slot_mapping"] = slot_mapping.numpy()
    block_idx = torch.div(slot_mapping, bs, rounding_mode="floor").tolist()
    block_off = (slot_mapping % bs).tolist()
    for i in range(T):
        kc[block_idx[i], block_off[i], :, :] = key_t[i]
        vc[block_idx[i], block_off[i], :, :] = val_t[i]
    out["cache_key_cache_out"], out["cache_value_cache_out"] = kc.numpy(), vc.numpy()
    # ---- greedy sampler front end (sampler.py:270-287, 473: argmax of the logprobs; temperature 0 -> 1.0)
    V = 4099
    logits = (torch.randn(6, V, generator=g) * 3).to(torch.float16)
    logits[1, 77] = logits[1].max() + 1           # a clear winner
    logits[2, 100] = logits[2, 3000] = logits[2].max() + 0.5   # an exact tie: argmax takes the first index
    logits[3] = 0                                 # uniform row
    logits[4, 5] = 60000                          # near the fp16 ceiling, everything else underflows
    lf = logits.to(torch.float)
    lf.div_(torch.ones(6).unsqueeze(1))
    probs = torch.softmax(lf, dim=-1, dtype=torch.float)
    logprobs = torch.log_softmax(lf, dim=-1, dtype=torch.float)
    out["sm_logits"] = logits.numpy()
    out["sm_probs"] = probs.numpy()
    out["sm_logprobs"] = logprobs.numpy()
    out["sm_argmax"] = torch.argmax(logprobs, dim=-1).numpy()
    np.savez_compressed(os.path.join(OUT, "rope_cache_softmax.npz"), **out)


def gen_sym_quant_w4a16():
    """a3 / a6 pins: the reference's OWN Python statements of the row-absmax quantiser and of the W4A16 linear,
    run here on CPU.
      * a3: third-party/QuaRot/quarot/nn/quantization.py:10 (the commented form that `fuse_sym_quant` replaced):
            scales = (max|x| / 7).to(fp16) * clip;  q = quarot.sym_quant(x, scales) = pack_i4(sym_quant(x, scale, 7)[0])
            -- quarot/functional/quantization.py:29-32 (`x / scale` on fp16 CPU tensors rounds the quotient to fp16,
            torch.round is ties-to-even, clamp to [-8, 7]) and :42-49 (pack_i4).  Same arithmetic as
            rowAbsMaxQuantizeKernel (quant.cu:102-167) for every row with a non-zero maximum and a clip ratio that is an
            fp16 value (the CUDA kernel rounds `clip` to fp16 before the product, Python multiplies by the fp32 scalar; an
            all-zero row is NaN in Python -- pack_i4's range assert refuses it -- and 0 in the kernel: those two cases stay
            pinned by the CUDA source alone).
      * a6: quarot_nn/linear.py:111-119 (commented statement of forward_w4a16): unpack_i4(weight).to(fp16) * scales ->
            fp16 weight, then the matmul; here with an fp32-accumulate matmul of the fp16 operands (what
            quarot_nn/qspec_gemm.py:20-88 accumulates in) and with torch's own fp16 CPU matmul."""
    qz = load_by_path("ref_quantization", f"{REF}/third-party/QuaRot/quarot/functional/quantization.py")
    g = torch.Generator().manual_seed(7)
    out = {}
    maxq = torch.tensor(7)
    cases = []
    for idx, (T, K, clip) in enumerate([(6, 256, 1.0), (5, 4096, 1.0), (4, 14336, 1.0), (4, 512, 0.875), (3, 256, 0.5)]):
        x = (torch.randn(T, K, generator=g) * (0.02 + 3.0 * torch.rand(T, 1, generator=g))).to(torch.float16)
        if idx == 0:
            x[0] = 0
            x[0, :8] = torch.tensor([3.5, 2.5, -2.5, -3.5, 0.5, -0.5, 1.5, 7.0])      # scale 1: exact .5 ties
            x[1, 0], x[1, 1] = 65504.0, -65504.0                                       # the fp16 ceiling
            x[2] = (x[2].float() * 1e-4).to(torch.float16)                             # subnormal-range row
            x[3, 5] = 40.0                                                             # one outlier: everything else -> 0 / +-1
        scales = (torch.max(torch.abs(x), dim=-1)[0].unsqueeze(1) / 7).to(torch.float16) * clip
        assert scales.dtype == torch.float16 and (scales > 0).all()
        q, _ = qz.sym_quant(x, scales, maxq)
        assert q.dtype == torch.float16
        packed = qz.pack_i4(q.to(torch.int8))
        key = f"sq{idx}_"
        out[key + "x"], out[key + "clip"] = x.numpy(), np.array(clip, np.float32)
        out[key + "scale"], out[key + "q"] = scales.view(-1).numpy(), packed.numpy().view(np.int8)
        cases.append((T, K, clip))
    out["sq_cases"] = np.array(len(cases))
    for idx, (M, N, K) in enumerate([(1, 64, 128), (4, 96, 4096), (16, 64, 4096), (3, 48, 14336), (16, 32, 13824)]):
        x = torch.randn(M, K, generator=g).to(torch.float16)
        wq = torch.randint(-8, 8, (N, K), generator=g, dtype=torch.int8)
        ws = (torch.rand(N, generator=g) * 0.01 + 0.001).to(torch.float16)
        packed = qz.pack_i4(wq)                                  # the checkpoint layout (uint8, low nibble = even k)
        unpacked = qz.unpack_i4(packed)
        assert torch.equal(unpacked.to(torch.int8), wq)
        w16 = qz.sym_dequant(unpacked.to(torch.float16), ws.view(-1, 1))      # linear.py:111-119: fp16 weight (one rounding)
        assert w16.dtype == torch.float16
        ref32 = (x.float() @ w16.float().t())                    # fp32 accumulate of the fp16 operands
        ref16 = (x @ w16.t())                                    # torch's fp16 CPU matmul (internal accumulation in fp32)
        key = f"wa{idx}_"
        out[key + "x"], out[key + "wq"], out[key + "ws"] = x.numpy(), packed.numpy().view(np.int8), ws.numpy()
        out[key + "ref_f32"], out[key + "ref_f16"] = ref32.numpy(), ref16.numpy()
    out["wa_cases"] = np.array(5)
    np.savez_compressed(os.path.join(OUT, "sym_quant_w4a16.npz"), **out)


def gen_typical_acceptance():
    """Run the reference TypicalAcceptanceSampler (vllm/model_executor/layers/typical_acceptance_sampler.py) on CPU: it is
    deterministic, so inputs + outputs are the whole fixture.  Flavours: random peaked distributions, draft = target argmax
    (all accepted), one-hot targets (entropy 0: threshold = min(eps, alpha)), near-uniform targets (high entropy: a tiny
    threshold), and thresholds / alphas across the reference's defaults (0.09 / 0.3, spec_decode_worker.py:95-110)."""
    import_ref_sampler()
    ta = load_by_path("vllm.model_executor.layers.typical_acceptance_sampler",
                      f"{REF}/vllm/model_executor/layers/typical_acceptance_sampler.py")
    out = {}
    g = torch.Generator().manual_seed(11)
    idx = 0
    for (B, k, V, thr, alpha) in [(4, 3, 257, 0.09, 0.3), (3, 5, 1000, 0.09, 0.3), (10, 5, 512, 0.3, 0.6), (1, 1, 64, 0.01, 0.1),
                                  (8, 4, 2048, 0.09, 0.3)]:
        for flavour in ("random", "argmax", "onehot", "flat"):
            if flavour == "random":
                tq = torch.softmax(torch.randn(B, k + 1, V, generator=g) * 4, -1)
                ids = torch.multinomial(tq[:, :k].reshape(-1, V), 1, generator=g).reshape(B, k)
                ids = torch.where(torch.rand(B, k, generator=g) < 0.5, ids, torch.randint(0, V, (B, k), generator=g))
            elif flavour == "argmax":
                tq = torch.softmax(torch.randn(B, k + 1, V, generator=g) * 6, -1)
                ids = tq[:, :k].argmax(-1)
            elif flavour == "onehot":
                tq = torch.zeros(B, k + 1, V)
                hot = torch.randint(0, V, (B, k + 1), generator=g)
                tq[torch.arange(B)[:, None], torch.arange(k + 1)[None], hot] = 1
                agree = torch.rand(B, k, generator=g) < 0.6
                ids = torch.where(agree, hot[:, :k], torch.randint(0, V, (B, k), generator=g))
            else:
                tq = torch.softmax(torch.randn(B, k + 1, V, generator=g) * 0.05, -1)
                ids = torch.randint(0, V, (B, k), generator=g)
            bonus = torch.randint(0, V, (B, 1), generator=g)
            sampler = ta.TypicalAcceptanceSampler(posterior_threshold=thr, posterior_alpha=alpha)
            sampler.init_tensors(device="cpu", device_type="cpu")
            o = sampler(tq, bonus, draft_probs=None, draft_token_ids=ids)
            key = f"t{idx}_"
            out[key + "tq"], out[key + "ids"], out[key + "bonus"] = tq.numpy().astype(np.float32), ids.numpy(), bonus.numpy()
            out[key + "params"] = np.array([thr, alpha], np.float64)
            out[key + "out"] = o.numpy()
            out[key + "accepted"] = sampler._evaluate_accepted_tokens(tq[:, :-1], ids).numpy()
            out[key + "counters"] = np.array([int(sampler.num_accepted_tokens), int(sampler.num_emitted_tokens),
                                              int(sampler.num_draft_tokens)], np.int64)
            idx += 1
    out["cases"] = np.array(idx)
    np.savez_compressed(os.path.join(OUT, "typical_acceptance.npz"), **out)


def _ref_function(path, *names):
    """Functions of a reference module that cannot be imported as a whole (vllm/model_executor/layers/sampler.py pulls in
    msgspec-backed vllm.sequence): the file is parsed, the named top-level function definitions are compiled FROM THE
    REFERENCE FILE and executed here with torch in scope.  Nothing is copied; the reference's own code runs."""
    import ast
    from typing import List, Optional
    tree = ast.parse(open(path).read(), filename=path)
    ns = {"torch": torch, "Optional": Optional, "List": List, "SequenceGroupToSample": object}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), ns)
    return [ns[n] for n in names]


def gen_sampling():
    """Non-greedy Sampler.forward (sampler.py:216-316) on CPU with the reference's own _apply_top_k_top_p (:387-413) and
    _multinomial (:585-604) executed from the reference file, temperature scaling and softmax as :266-287; the exponential
    draws of _multinomial are recorded (torch.Tensor.exponential_ replaced by a recorded array, as for the rejection sampler).
    Rows of DISTINCT fp16 logits (no ties: the reference's top-p rule then has no sort-order ambiguity) and, separately, rows
    with many equal logits (ties: compared outside the boundary group only)."""
    apply_tk_tp, multinomial = _ref_function(f"{REF}/vllm/model_executor/layers/sampler.py", "_apply_top_k_top_p", "_multinomial")
    g = torch.Generator().manual_seed(13)
    out = {}
    V = 6007
    allh = torch.arange(-2 ** 15, 2 ** 15, dtype=torch.int32).to(torch.int16).view(torch.float16)
    pool = allh[torch.isfinite(allh) & (allh.abs() < 12) & (allh.abs() > 1e-3)]
    pool = torch.unique(pool)
    cases = [(1.0, -1, 1.0), (0.7, 50, 1.0), (1.3, -1, 0.9), (0.8, 40, 0.95), (1.0, 1, 1.0), (2.0, 5000, 0.5), (0.5, -1, 0.3),
             (1.0, 7, 0.01), (0.9, 6007, 0.999)]
    T = len(cases)
    for name, ties in (("distinct", False), ("ties", True)):
        if ties:
            logits = (torch.randn(T, V, generator=g) * 2).to(torch.float16)
            logits = (logits * 4).round() / 4                      # quarter steps: dozens of equal logits everywhere
            logits = logits.to(torch.float16)
        else:
            logits = torch.stack([pool[torch.randperm(pool.numel(), generator=g)[:V]] for _ in range(T)])
            # a peaked shape: scale a few so that top-p cuts somewhere interesting
        temperature = torch.tensor([c[0] for c in cases], dtype=torch.float32)
        top_k = torch.tensor([V if c[1] == -1 else c[1] for c in cases], dtype=torch.int32)   # vLLM maps -1 to vocab_size
        top_p = torch.tensor([c[2] for c in cases], dtype=torch.float32)
        lf = logits.to(torch.float)
        lf.div_(temperature.unsqueeze(dim=1))
        masked = apply_tk_tp(lf.clone(), top_p, top_k)
        probs = torch.softmax(masked, dim=-1, dtype=torch.float)
        E = torch.empty(T, V).exponential_(1.0, generator=g)
        orig = torch.Tensor.exponential_
        try:
            torch.Tensor.exponential_ = lambda self, *a, **kw: self.copy_(E)
            tok = multinomial(probs.clone(), 1).view(-1)
        finally:
            torch.Tensor.exponential_ = orig
        out[name + "_logits"] = logits.numpy()
        out[name + "_temperature"], out[name + "_top_k"], out[name + "_top_p"] = temperature.numpy(), top_k.numpy(), top_p.numpy()
        out[name + "_E"] = E.numpy()
        out[name + "_keep"] = torch.isfinite(masked).numpy()
        out[name + "_probs"] = probs.numpy()
        out[name + "_token"] = tok.numpy()
    np.savez_compressed(os.path.join(OUT, "sampling.npz"), **out)


if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference checkout is only present in the build container"
    gen_pack()
    gen_hadamard()
    gen_w4a4()
    gen_rejection()
    gen_attention()
    gen_rope_cache_softmax()
    gen_sym_quant_w4a16()
    gen_typical_acceptance()
    gen_sampling()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
