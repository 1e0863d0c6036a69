"""vLLM quantization plugin surface for the QSpec weights.

    QuantizationConfig   vllm/model_executor/layers/quantization/base_config.py:58-135   (6 abstract methods)
    LinearMethodBase     vllm/model_executor/layers/linear.py:85-116                     (create_weights / apply)
    registration         vllm/model_executor/layers/quantization/__init__.py:36-69       (register_quantization_config)

The reference model builds `quarot_nn.Linear4bit` modules directly (quarot_llama.py:152-173,301-314); the same
weights can also be reached the vLLM-native way -- a `*ParallelLinear` layer asking its quantization config for a
linear method -- and this module is that path: `QSpecConfig.get_quant_method(layer, prefix)` returns a
`QSpecLinearMethod`, whose `create_weights` registers the parameters under the names of the reference checkpoint
(`weight` [N, K/2] packed int4, `weight_scales` [N, 1] fp16; quarot_nn/linear.py:28-63) and whose `apply` runs either
view of the ONE packed buffer: a `PackedQuantizedTensor` input takes the W4A4 GEMM (linear.py:67-84), an fp16 tensor
the W4A16 GEMM (linear.py:102-124).  Which one arrives is decided by the caller's `w4a4` flag exactly as in the
reference: the Quantizer in front of the layer either ran or did not.

vllm is imported if it is importable (then the config is registered as "qspec"); otherwise structurally identical
stand-ins of the two base classes are used so that the mirror can be exercised without vllm.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Dict, List, Optional

import torch

from . import ops
from .quarot_nn import PackedQuantizedTensor

try:  # pragma: no cover - vllm is absent from the build image
    from vllm.model_executor.layers.linear import LinearBase, LinearMethodBase
    from vllm.model_executor.layers.quantization import register_quantization_config
    from vllm.model_executor.layers.quantization.base_config import QuantizationConfig
    from vllm.model_executor.utils import set_weight_attrs
    HAVE_VLLM = True
except Exception:  # ModuleNotFoundError (msgspec, ...) in this image
    HAVE_VLLM = False

    class QuantizationConfig(ABC):   # base_config.py:58-135, abstract surface only
        @abstractmethod
        def get_name(self) -> str: ...

        @abstractmethod
        def get_supported_act_dtypes(self) -> List[torch.dtype]: ...

        @classmethod
        @abstractmethod
        def get_min_capability(cls) -> int: ...

        @staticmethod
        @abstractmethod
        def get_config_filenames() -> List[str]: ...

        @classmethod
        @abstractmethod
        def from_config(cls, config: Dict[str, Any]) -> "QuantizationConfig": ...

        @abstractmethod
        def get_quant_method(self, layer: torch.nn.Module, prefix: str): ...

        @staticmethod
        def get_from_keys(config: Dict[str, Any], keys: List[str]) -> Any:
            """First of `keys` present in `config` (same contract as the vLLM helper: ValueError when none is)."""
            hit = next((k for k in keys if k in config), None)
            if hit is None:
                raise ValueError(f"none of {keys} is in the quantization config")
            return config[hit]

        @staticmethod
        def get_from_keys_or(config: Dict[str, Any], keys: List[str], default: Any) -> Any:
            hit = next((k for k in keys if k in config), None)
            return default if hit is None else config[hit]

    class LinearMethodBase(ABC):     # linear.py:85-116
        @abstractmethod
        def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                           params_dtype, **extra_weight_attrs): ...

        @abstractmethod
        def apply(self, layer, x, bias=None): ...

    LinearBase = torch.nn.Module

    def set_weight_attrs(weight: torch.Tensor, weight_attrs: Optional[Dict[str, Any]]):
        """Attach loader metadata to a parameter (vllm.model_executor.utils.set_weight_attrs): no silent overwrite."""
        for key, value in (weight_attrs or {}).items():
            if hasattr(weight, key):
                raise AssertionError(f"parameter attribute {key!r} is already set")
            setattr(weight, key, value)

    def register_quantization_config(name: str):
        def _wrap(cls):
            return cls
        return _wrap


class QSpecLinearMethod(LinearMethodBase):
    """Linear method over packed int4 weights with per-output-channel fp16 scales (both QSpec views)."""

    def __init__(self, quant_config: "QSpecConfig"):
        self.quant_config = quant_config

    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int, output_partition_sizes: List[int],
                       input_size: int, output_size: int, params_dtype: torch.dtype, **extra_weight_attrs):
        if params_dtype != torch.float16:
            raise ValueError("QSpec weights carry fp16 scales; params_dtype must be torch.float16")
        if input_size_per_partition % 128:
            raise ValueError(f"input_size_per_partition={input_size_per_partition} must be a multiple of 128 "
                             "(one K step of the int4 GEMMs; a row-parallel shard must keep whole steps)")
        out = sum(output_partition_sizes)
        if out % 16:
            raise ValueError(f"output partition {out} must be a multiple of 16 (one weight tile)")
        # names and layouts of quarot_nn.Linear4bit (linear.py:28-63): fused layers stack their logical weights on dim 0
        weight = torch.nn.Parameter(torch.zeros(out, input_size_per_partition // 2, dtype=torch.int8), requires_grad=False)
        set_weight_attrs(weight, {"input_dim": 1, "output_dim": 0, "packed_dim": 1, "pack_factor": 2})
        scales = torch.nn.Parameter(torch.zeros(out, 1, dtype=params_dtype), requires_grad=False)
        set_weight_attrs(scales, {"output_dim": 0})
        layer.register_parameter("weight", weight)
        layer.register_parameter("weight_scales", scales)
        set_weight_attrs(weight, extra_weight_attrs)
        set_weight_attrs(scales, extra_weight_attrs)
        layer.logical_widths = list(output_partition_sizes)

    def apply(self, layer: torch.nn.Module, x, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
        """x: PackedQuantizedTensor (draft pass, W4A4) or fp16 [..., K] (verify pass, W4A16)."""
        w, s = layer.weight, layer.weight_scales.view(-1)
        N = w.shape[0]
        if isinstance(x, PackedQuantizedTensor):
            xq, xs = x.quantized_x, x.scales_x
            lead = xq.shape[:-1]
            xq2 = xq.reshape(-1, xq.shape[-1])
            C = out if out is not None else torch.empty(xq2.shape[0], N, dtype=torch.float16, device=xq.device)
            ops.rowwise_scaled_linear_cutlass_s4s4_unified(xq2, xs.reshape(-1), w, s, bias, C)
            return C.view(*lead, N)
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        C = out if out is not None else torch.empty(x2.shape[0], N, dtype=torch.float16, device=x.device)
        ops.w4a16_linear(x2, w, s, C, bias)
        return C.view(*lead, N)


@register_quantization_config("qspec")
class QSpecConfig(QuantizationConfig):
    """`quantization="qspec"`: 4-bit symmetric per-channel weights shared by a W4A4 draft and a W4A16 verify pass."""

    def __init__(self, weight_bits: int = 4, act_bits_draft: int = 4, clip_ratio: float = 1.0):
        if weight_bits != 4 or act_bits_draft != 4:
            raise ValueError("QSpec is W4A4-draft / W4A16-verify: weight_bits and act_bits_draft must be 4")
        self.weight_bits, self.act_bits_draft, self.clip_ratio = weight_bits, act_bits_draft, clip_ratio

    def __repr__(self) -> str:
        return f"QSpecConfig(weight_bits=4, draft=W4A4, verify=W4A16, clip_ratio={self.clip_ratio})"

    def get_name(self) -> str:
        return "qspec"

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        return [torch.float16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 90   # gfx9 family as vLLM's ROCm platform reports it (major * 10 + minor); the kernels are gfx950

    @staticmethod
    def get_config_filenames() -> List[str]:
        return ["quantize_config.json", "quant_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "QSpecConfig":
        wb = cls.get_from_keys_or(config, ["w_bits", "weight_bits", "bits"], 4)
        ab = cls.get_from_keys_or(config, ["a_bits", "act_bits_draft"], 4)
        clip = cls.get_from_keys_or(config, ["a_clip_ratio", "clip_ratio"], 1.0)
        return cls(int(wb), int(ab), float(clip))

    def get_quant_method(self, layer: torch.nn.Module, prefix: str) -> Optional[QSpecLinearMethod]:
        # every decoder linear is quantised; embed_tokens / lm_head are plain fp16 modules in the reference
        # (quarot_llama.py:470-480, 586-590) and never ask
        if HAVE_VLLM and not isinstance(layer, LinearBase):
            return None
        if prefix.endswith("lm_head") or "embed_tokens" in prefix:
            return None
        return QSpecLinearMethod(self)

    def get_scaled_act_names(self) -> List[str]:
        return []
