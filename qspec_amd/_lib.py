"""ctypes binding of libqspec_hip.so (the C ABI declared in include/qspec_hip.h).

There is no CPU fallback: if the HIP library is missing or a call fails, this
module raises.  The CPU oracle under ``oracle/`` is test infrastructure and is
never imported from here.
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QSPEC_HIP_LIB") or os.path.join(_HERE, "csrc", "libqspec_hip.so")  # env: instrumented dev builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "qspec_hip.h")

_vp, _i, _f, _i64, _u64, _sz = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64, ctypes.c_uint64,
                                ctypes.c_size_t)

# name -> (restype, argtypes); mirrors include/qspec_hip.h one to one
SIGNATURES = {
    "qspec_abi_version": (_i, []),
    "qspec_last_error": (ctypes.c_char_p, []),
    "qspec_rms_norm_general_fuse_sum_i4": (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _vp]),
    "qspec_rms_norm_general_fuse_sum_fp16": (_i, [_vp, _vp, _f, _i, _i, _vp]),
    "qspec_add_rms_norm_i4": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp]),
    "qspec_add_rms_norm_fp16": (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _vp]),
    "qspec_fuse_sym_quant": (_i, [_vp, _vp, _vp, _f, _i, _i, _vp]),
    "qspec_fast_hadamard_transform": (_i, [_vp, _f, _vp, _i64, _i, _vp]),
    "qspec_hadamard_mix": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "qspec_heads_hadamard": (_i, [_vp, _vp, _vp, _vp, _f, _f, _i, _i, _i, _vp]),
    "qspec_heads_hadamard_mix": (_i, [_vp, _vp, _vp, _f, _i, _i, _i, _i, _vp]),
    "qspec_silu_mul": (_i, [_vp, _vp, _i, _i, _vp]),
    "qspec_silu_mul_hadamard": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _i, _vp]),
    "qspec_xwg_workspace_bytes": (_sz, []),
    "qspec_mlp_hadamard": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _i, _vp, _vp]),
    "qspec_qkv_rope_linear_s4s4": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "qspec_qkv_rope_linear_w4a16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp,
                                         _vp]),
    "qspec_qkv_rope_linear_supported": (_i, [_i, _i, _i, _i, _i]),
    "qspec_gate_up_silu_linear_s4s4": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "qspec_gate_up_silu_linear_w4a16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "qspec_ln_qkv_rope_linear_s4s4": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i,
                                           _i, _i, _vp, _vp]),
    "qspec_ln_gate_up_silu_linear_s4s4": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "qspec_ln_linear_workspace_bytes": (_sz, []),
    "qspec_ln_linear_s4s4_supported": (_i, [_i, _i, _i]),
    "qspec_rowwise_scaled_linear_s4s4_residual": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "qspec_rowwise_scaled_linear_s4s4_residual_supported": (_i, [_i, _i, _i]),
    "qspec_rowwise_scaled_linear_s4s4_residual_hq": (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "qspec_rowwise_scaled_linear_s4s4_residual_hq_supported": (_i, [_i, _i, _i, _i]),
    "qspec_heads_hadamard_merged_spread": (_i, [_vp, _i, _i, _vp, _vp, _f, _i, _i, _i, _vp]),
    "qspec_heads_hadamard_merged_spread_supported": (_i, [_i, _i, _i]),
    "qspec_heads_hadamard_mix_merged_spread": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _f, _i, _i, _i, _vp]),
    "qspec_heads_hadamard_mix_merged_spread_supported": (_i, [_i, _i, _i, _i]),
    "qspec_rowwise_scaled_linear_s4s4_partial_slices": (_i, [_i, _i, _i]),
    "qspec_rowwise_scaled_linear_s4s4_partial": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "qspec_add_rms_norm_ipartial": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _i, _i, _vp]),
    "qspec_rowwise_scaled_linear_s4s4": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "qspec_w4a16_workspace_bytes": (_sz, []),
    "qspec_w4a16_linear": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "qspec_w4a16_linear_partial_slices": (_i, [_i, _i, _i]),
    "qspec_w4a16_linear_partial": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "qspec_add_rms_norm_fp16_partial": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _f, _i, _i, _vp]),
    "qspec_w4a16_linear_ksliced": (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "qspec_w4a16_linear_ksliced_raw": (_i, [_vp, _i64, _vp, _i64, _vp, _i, _i, _i, _vp, _vp]),
    "qspec_gate_up_silu_linear_w4a16_shard": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "qspec_oneshot_create": (_i, [_i, _i, _sz, ctypes.POINTER(_vp)]),
    "qspec_oneshot_handle_bytes": (_i, []),
    "qspec_oneshot_local_handle": (_i, [_vp, _vp]),
    "qspec_oneshot_open_peers": (_i, [_vp, _vp]),
    "qspec_oneshot_all_reduce_f32": (_i, [_vp, _vp, _i, _vp]),
    "qspec_oneshot_error": (_i, [_vp]),
    "qspec_oneshot_error_word": (_vp, [_vp]),
    "qspec_oneshot_destroy": (_i, [_vp]),
    "qspec_linear_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "qspec_dequant_w4": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "qspec_rotary_embedding": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i64, _i64, _vp]),
    "qspec_reshape_and_cache_flash": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _vp]),
    "qspec_rope_kv_write": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "qspec_heads_hadamard_merged": (_i, [_vp, _i, _i, _vp, _vp, _vp, _f, _f, _i, _i, _i, _vp]),
    "qspec_paged_attention_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "qspec_paged_attention": (_i, [_vp, _i64, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp,
                                   _vp, _vp]),
    "qspec_embedding": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "qspec_sampler_workspace_bytes": (_sz, [_i]),
    "qspec_softmax_argmax": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "qspec_lm_head_sampler_workspace_bytes": (_sz, [_i]),
    "qspec_lm_head_softmax_argmax_supported": (_i, [_i, _i, _i]),
    "qspec_lm_head_softmax_argmax": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "qspec_rejection_sample": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _u64, _u64, _vp, _i, _i, _i, _i64, _i64, _i64, _i64,
                                    _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qspec_sample_workspace_bytes": (_sz, [_i]),
    "qspec_sample_top_k_top_p": (_i, [_vp, _vp, _vp, _vp, _vp, _u64, _u64, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp]),
    "qspec_typical_acceptance_sample": (_i, [_vp, _vp, _vp, _f, _f, _i, _i, _i, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qspec_advance_step_flashattn": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "qspec_spec_prepare_draft": (_i, [_i, _i, _i, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "qspec_spec_advance_draft": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "qspec_spec_prepare_verify": (_i, [_i, _i, _i, _i, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "qspec_spec_prepare_draft_embed": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "qspec_spec_advance_draft_embed": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i, _i, _vp]),
    "qspec_spec_prepare_verify_embed": (_i, [_i, _i, _i, _i, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "qspec_spec_commit": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "qspec_spec_snapshot": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qspec_collect_error_words": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
}

# Fragment-major activation tiles (include/qspec_hip.h, "activation layout"): each of these has an `_xp` twin with the same
# prototype, reading / writing the 16-row tile in MFMA operand order instead of row-major.
XP_TWINS = ("qspec_add_rms_norm_fp16", "qspec_add_rms_norm_fp16_partial", "qspec_heads_hadamard_merged",
            "qspec_heads_hadamard_mix_merged_spread", "qspec_mlp_hadamard", "qspec_w4a16_linear",
            "qspec_w4a16_linear_partial", "qspec_qkv_rope_linear_w4a16", "qspec_gate_up_silu_linear_w4a16")
SIGNATURES.update({name + "_xp": SIGNATURES[name] for name in XP_TWINS})
SIGNATURES["qspec_w4a16_act_layout_supported"] = (_i, [_i, _i])
SIGNATURES["qspec_w4a16_act_layout32_supported"] = (_i, [_i, _i, _i])
SIGNATURES["qspec_w4a16_linear_xp32"] = (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp])
SIGNATURES["qspec_w4a16_linear_partial_slices_xp32"] = (_i, [_i, _i, _i])
SIGNATURES["qspec_w4a16_linear_partial_xp32"] = (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp])
SIGNATURES["qspec_qkv_rope_linear_w4a16_xp32"] = (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp])
SIGNATURES["qspec_gate_up_silu_linear_w4a16_xp32"] = (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp])
SIGNATURES["qspec_mlp_hadamard_act_layout_supported"] = (_i, [_i, _i, _i])


# entry points only the experimental build exports (csrc/experimental/qspec_hip_experimental.h): bound when present
EXPERIMENTAL_SIGNATURES = {
    "qspec_prefetch": (_i, [_vp, _sz, _i, _vp]),
    "qspec_prefetch_tiles": (_i, [_vp, _sz, _i, _i, _i, _vp]),
}
EXPERIMENTAL_LIB_PATH = os.path.join(_HERE, "csrc", "libqspec_hip_experimental.so")


def header_symbols(path: str = HEADER_PATH):
    """Every function name declared in include/qspec_hip.h."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qspec_[a-z0-9_]+)\s*\(", text)))


class QSpecLibraryError(RuntimeError):
    pass


_lib = None


def load():
    """Load libqspec_hip.so, binding every prototype.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QSpecLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C qspec_amd/csrc`).  There is no CPU fallback for the QSpec hot path.")
    # torch first: its bundled HIP runtime must be THE runtime of the process (device memory and streams come from
    # torch).  Loaded the other way round, this library binds to the system libamdhip64 and a later torch brings a
    # second runtime: every launch then fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in EXPERIMENTAL_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, lib=None):
    if rc != 0:
        lib = lib or load()
        raise QSpecLibraryError(lib.qspec_last_error().decode())
