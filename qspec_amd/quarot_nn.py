"""Host-side mirror of the reference's operator package
`vllm/model_executor/layers/quarot_nn/{linear,normalization,quantization,hadamard}.py`
(+ `quarot.PackedQuantizedTensor`, third-party/QuaRot/quarot/__init__.py:154-170).

Same class names, constructor arguments, forward signatures and kwargs
(`w4a4`, scratch-buffer names of vllm/spec_decode/draft_model_runner.py:311-320),
so code written against the reference modules runs against these.  Each call is
one HIP kernel from libqspec_hip.so (the reference needs 1-4 launches per
call plus transposes / a full-weight XOR pass).  `w4a8` is not on the QSpec
path and raises.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import hadamard_tables, ops


class PackedQuantizedTensor:
    """quarot.PackedQuantizedTensor (third-party/QuaRot/quarot/__init__.py:154-170)."""

    def __init__(self, quantized_x: torch.Tensor, scales_x: torch.Tensor):
        self.quantized_x = quantized_x
        self.scales_x = scales_x

    def size(self):
        return self.quantized_x.size()

    @property
    def device(self):
        return self.quantized_x.device

    @property
    def dtype(self):
        return self.quantized_x.dtype


def _no_w4a8(kwargs):
    if kwargs.get("w4a8", False):
        raise NotImplementedError("w4a8 is never enabled on the QSpec path (quarot_nn/linear.py:86, SURVEY.md 2b)")


class Linear4bit(torch.nn.Module):
    """quarot_nn/linear.py:28-124.  `weight` uint8/int8 [N, K/2] (lo nibble = even k), `weight_scales` fp16 [N,1].
    Draft and verify read the SAME `weight` buffer; there is no `weight ^ mask` copy (linear.py:122)."""

    def __init__(self, in_features, out_features, bias=False, dtype=torch.float16, device=None, **kwargs):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.register_buffer("weight_scales", torch.zeros((out_features, 1), dtype=dtype, device=device))
        self.register_buffer("weight", torch.zeros((out_features, in_features // 2), dtype=torch.int8, device=device))
        if bias:
            self.register_buffer("bias", torch.zeros((out_features,), dtype=dtype, device=device))
        else:
            self.bias = None

    def forward(self, x, C=None, **kwargs):
        _no_w4a8(kwargs)
        if kwargs.get("w4a4", False):
            return self.forward_w4a4(x, C)
        return self.forward_w4a16(x, C)

    def _scales(self):
        return self.weight_scales.view(-1)

    def forward_w4a4(self, x, C=None):
        assert type(x) == PackedQuantizedTensor  # quantized input is given (linear.py:72)
        xq, xs = x.quantized_x, x.scales_x
        if C is None:
            C = torch.empty(xq.shape[0], self.out_features, dtype=torch.float16, device=xq.device)
        return ops.rowwise_scaled_linear_cutlass_s4s4_unified(xq, xs, self.weight, self._scales(), self.bias, C)

    def forward_w4a16(self, x, C=None):
        if C is None:
            C = torch.empty(x.shape[0], self.out_features, dtype=torch.float16, device=x.device)
        return ops.w4a16_linear(x, self.weight, self._scales(), C, self.bias)

    @staticmethod
    def from_float(module: torch.nn.Linear, weight_scales=None, **kwargs):
        """linear.py:131-152: int = round(W / scale) packed two per byte."""
        w = module.weight.data
        m = Linear4bit(module.in_features, module.out_features, bias=module.bias is not None, dtype=w.dtype,
                       device=w.device, **kwargs)
        if weight_scales is not None:
            assert weight_scales.shape == (module.out_features, 1)
            m.weight_scales.copy_(weight_scales.to(w.dtype))
            q = (w / weight_scales.to(w.device)).round().clamp(-8, 7).to(torch.int8)
            m.weight.copy_(pack_i4(q))
            if module.bias is not None:
                m.bias.copy_(module.bias)
        return m


def pack_i4(q: torch.Tensor) -> torch.Tensor:
    """quarot.functional.pack_i4 (third-party/QuaRot/quarot/functional/quantization.py:42-49), int8 result."""
    assert q.dtype == torch.int8
    lo = q[..., 0::2] & 0xF
    hi = (q[..., 1::2] & 0xF) << 4
    return (lo | hi).to(torch.int8)


class RMSNorm(torch.nn.Module):
    """quarot_nn/normalization.py:5-105 -- actually LayerNorm without gamma (mean-subtracting)."""

    def __init__(self, mean_dim: int, eps=1e-5, fuse=False):
        super().__init__()
        self.eps = eps
        self.mean_dim = mean_dim
        self.fuse = fuse

    def forward(self, x, **kwargs):
        _no_w4a8(kwargs)
        if kwargs.get("w4a4", False):
            return self.fuse_forward(x, kwargs.get("quantized_buffer_qkv"), kwargs.get("scale_buffer"),
                                     kwargs.get("input_sum_buffer"))
        return self.unfuse_forward(x)

    def fuse_forward(self, x, out_q=None, scaling_factor=None, input_sum=None):
        T = x.numel() // x.shape[-1]
        H = x.shape[-1]
        if out_q is None or input_sum is None or scaling_factor is None:
            out_q = torch.empty(T, H // 2, dtype=torch.int8, device=x.device)
            both = torch.empty(2, T, dtype=torch.float16, device=x.device)
            input_sum, scaling_factor = both[0], both[1]
        ops.rms_norm_general_fuse_sum_i4(out_q, x, input_sum, scaling_factor, self.eps, True)
        return PackedQuantizedTensor(out_q, scaling_factor)

    def unfuse_forward(self, x, out=None):
        if out is None:
            out = torch.empty_like(x)
        ops.rms_norm_general_fuse_sum_fp16(out, x, self.eps)
        return out


class Quantizer(torch.nn.Module):
    """quarot_nn/quantization.py:4-30: draft -> row abs-max int4; verify -> identity."""

    def __init__(self, input_clip_ratio=1.0, **kwargs):
        super().__init__()
        self.input_clip_ratio = input_clip_ratio

    def forward(self, x, scale=None, out=None, **kwargs):
        _no_w4a8(kwargs)
        if kwargs.get("w4a4", False):
            T, K = x.shape
            if scale is None or out is None:
                scale = torch.empty(T, dtype=torch.float16, device=x.device)
                out = torch.empty(T, K // 2, dtype=torch.int8, device=x.device)
            ops.fuse_sym_quant(x, scale, out, self.input_clip_ratio)
            return PackedQuantizedTensor(out, scale)
        return x


class OnlineHadamard(torch.nn.Module):
    """quarot_nn/hadamard.py:7-41.  The reference hard-codes 1/sqrt(32) and 1/sqrt(14336) (:12-13); here the
    scale is 1/sqrt(hadamard_dim) computed the same way (`1.0/torch.tensor(n).sqrt()`, an fp32 value), which
    is identical for Llama-3-8B and generalises to the other configs."""

    def __init__(self, hadamard_dim, force_fp32=False, device=None):
        super().__init__()
        if force_fp32:
            raise NotImplementedError("fp32 Hadamard is not used on the QSpec path")
        self.hadamard_dim = hadamard_dim
        had, self.rem_dim = hadamard_tables.get_hadK(hadamard_dim)
        self.scale = float(1.0 / torch.tensor(hadamard_dim).sqrt())
        if had is not None:
            self.register_buffer("had_rem_dim", had.to(torch.float16).to(device))
        else:
            self.had_rem_dim = None

    def forward(self, x, out=None, **kwargs):
        """x [..., n]; transforms the last dim with (hadK (x) H_{n/K}) / sqrt(n) (matmul_hadU_cuda,
        quarot/functional/hadamard.py:113-122)."""
        _no_w4a8(kwargs)
        n = x.shape[-1]
        K = self.rem_dim
        x2 = x.reshape(-1, n // K)
        y = ops.faster_fast_hadamard_transform(x2.contiguous(), self.scale,
                                               out if out is not None and out.shape == x2.shape else torch.empty_like(x2))
        if K == 1:
            return y.view(x.shape)
        z = torch.empty(x.numel() // n, K, n // K, dtype=x.dtype, device=x.device)
        ops.hadamard_mix(y.view(-1, K, n // K), self.had_rem_dim, z)
        return z.view(x.shape)
