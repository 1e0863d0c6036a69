"""`torch.ops.qspec.*`: the operator boundary of SURVEY.md 8b.2 registered with the PyTorch dispatcher.

The reference reaches its kernels as `torch.ops.torchao.rowwise_scaled_linear_cutlass_s4s4_unified`
(third-party/ao/torchao/ops.py:29), `layernorm_ops.rms_norm_general_fuse_sum_{i4,fp16}` (pybind,
quarot_nn/normalization.py:56-80), `quarot._CUDA.fuse_sym_quant` (quarot/__init__.py:119-144),
`fast_hadamard_transform_cuda.*` (quarot/functional/hadamard.py:94-124), `bitblas.Matmul` (quarot_nn/linear.py:122)
and the stock vLLM `_C` / `_C_cache_ops` entries.  Importing this module defines the same operators, same argument
order, under one namespace, implemented by `qspec_amd.ops` (ctypes -> libqspec_hip.so) for the CUDA/HIP dispatch key
ONLY: a CPU tensor gets the dispatcher's "no kernel for backend CPU" error -- there is no fallback.
"""
from __future__ import annotations

import torch

from . import ops

_LIB = torch.library.Library("qspec", "DEF")


def _op(schema: str, fn):
    _LIB.define(schema)
    _LIB.impl(schema.split("(")[0], fn, "CUDA")


def _none(fn):
    def wrapper(*a, **k):
        fn(*a, **k)
    return wrapper


_op("rms_norm_general_fuse_sum_i4(Tensor(a!) out_q, Tensor x, Tensor(b!)? input_sum, Tensor(c!) scale, float eps, "
    "bool use_per_token_quant=True) -> ()", _none(ops.rms_norm_general_fuse_sum_i4))
_op("rms_norm_general_fuse_sum_fp16(Tensor(a!) out, Tensor x, float eps) -> ()", _none(ops.rms_norm_general_fuse_sum_fp16))
_op("fuse_sym_quant(Tensor x, Tensor(a!) scale, Tensor(b!) q, float clip_ratio=1.0) -> ()", _none(ops.fuse_sym_quant))
_op("faster_fast_hadamard_transform(Tensor x, float scale, Tensor(a!) out) -> Tensor(a!)", ops.faster_fast_hadamard_transform)
_op("fast_hadamard_transform(Tensor x, float scale=1.0) -> Tensor", ops.fast_hadamard_transform)
_op("rowwise_scaled_linear_cutlass_s4s4_unified(Tensor xq, Tensor x_scale, Tensor wq, Tensor w_scale, Tensor? bias, "
    "Tensor(a!) out) -> Tensor(a!)", ops.rowwise_scaled_linear_cutlass_s4s4_unified)
# bitblas.Matmul.__call__(x, w, output=, scale=, bias=): the same packed buffer as the s4s4 op, no XOR copy
_op("w4a16_matmul(Tensor x, Tensor wq, Tensor(a!) output, Tensor scale, Tensor? bias=None) -> Tensor(a!)",
    lambda x, wq, output, scale, bias=None: ops.w4a16_linear(x, wq, scale.reshape(-1), output, bias))
_op("rotary_embedding(Tensor positions, Tensor(a!) query, Tensor(b!) key, int head_size, Tensor cos_sin_cache, "
    "bool is_neox=True) -> ()", _none(ops.rotary_embedding))
_op("reshape_and_cache_flash(Tensor key, Tensor value, Tensor(a!) key_cache, Tensor(b!) value_cache, Tensor slot_mapping) -> ()",
    _none(ops.reshape_and_cache_flash))
_op("advance_step_flashattn(int num_seqs, int num_queries, int block_size, Tensor(a!) input_tokens, Tensor sampled_token_ids, "
    "Tensor(b!) input_positions, Tensor(c!) seq_lens, Tensor(d!) slot_mapping, Tensor block_tables) -> ()",
    _none(ops.advance_step_flashattn))

OPS = ("rms_norm_general_fuse_sum_i4", "rms_norm_general_fuse_sum_fp16", "fuse_sym_quant", "faster_fast_hadamard_transform",
       "fast_hadamard_transform", "rowwise_scaled_linear_cutlass_s4s4_unified", "w4a16_matmul", "rotary_embedding",
       "reshape_and_cache_flash", "advance_step_flashattn")
