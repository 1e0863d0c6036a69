"""Generate the committed golden fixtures by RUNNING the importable pieces of the
reference (`/root/reference`, present only in the build container) on CPU.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz

What is imported from the reference (by file path; nothing is copied):
  * third-party/QuaRot/quarot/functional/quantization.py   pack_i4 / unpack_i4
  * third-party/QuaRot/quarot/functional/hadamard.py        get_hadK tables, matmul_hadU
      (needs the names `fast_hadamard_transform` / `fast_hadamard_transform_cuda` to exist at
       import time; they are CUDA extension modules absent here, so two EMPTY module objects are
       registered -- none of the functions called below touch them)
  * vllm/model_executor/layers/{spec_decode_base_sampler,rejection_sampler}.py   RejectionSampler on CPU
      (imported under an empty package shell `vllm` providing only envs.VLLM_USE_FLASHINFER_SAMPLER,
       logger.init_logger and platforms.current_platform.simple_compile_backend = "eager")
  * third-party/ao/test/test_rowwise_scaled_linear_cutlass.py:64-84 is a formula, restated below
    with torch CPU ops on inputs of the shapes that test lists (:14-21).

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


if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference checkout is only present in the build container"
    gen_pack()
    gen_hadamard()
    gen_w4a4()
    gen_rejection()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
