"""Tensor-level bindings: the operator boundary of the QSpec path.

Each function mirrors the call signature of the reference extension op it
replaces (names in the docstrings, paths relative to the reference checkout),
takes caller-owned contiguous HIP tensors, enqueues on the current torch stream
and never synchronises.  PyTorch is used for memory and streams only.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib

_F16, _I8, _U8, _I64, _I32, _F32 = torch.float16, torch.int8, torch.uint8, torch.int64, torch.int32, torch.float32


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ws_key(device):
    """Key of the process-wide scratch / hand-off workspaces: they carry ticket, flag or generation state and are
    safe for ONE stream at a time (launches on a stream are serialised), so each (device, stream) gets its own."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return (device.type, idx, torch.cuda.current_stream(device).cuda_stream)


def _chk(t: torch.Tensor, name: str, dtype=None):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a HIP/CUDA tensor (the QSpec hot path has no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if dtype is not None:
        ok = t.dtype in dtype if isinstance(dtype, tuple) else t.dtype == dtype
        if not ok:
            raise RuntimeError(f"{name} has dtype {t.dtype}, expected {dtype}")
    return t.data_ptr()


def _opt(t: Optional[torch.Tensor], name: str, dtype=None):
    return None if t is None else _chk(t, name, dtype)


def _call(name: str, *args):
    lib = _lib.load()
    _lib.check(getattr(lib, name)(*args), lib)


# ------------------------------------------------------------------ norm + quant

def rms_norm_general_fuse_sum_i4(out_q, x, input_sum, scaling, eps: float, use_per_token_quant: bool = True):
    """qserve_backend.layernorm_ops.rms_norm_general_fuse_sum_i4 (third-party/kernels/csrc/layernorm.cpp:80-83)."""
    if not use_per_token_quant:
        raise RuntimeError("only per-token quantisation exists in the reference launcher (layernorm_kernels.cu:903)")
    H = x.shape[-1]
    T = x.numel() // H
    _call("qspec_rms_norm_general_fuse_sum_i4", _chk(out_q, "out_q", _I8), _chk(x, "x", _F16),
          _opt(input_sum, "input_sum", _F16), _chk(scaling, "scaling", _F16), float(eps), T, H, _stream())


def rms_norm_general_fuse_sum_fp16(out, x, eps: float):
    """layernorm_ops.rms_norm_general_fuse_sum_fp16 (layernorm.cpp:85-87)."""
    H = x.shape[-1]
    T = x.numel() // H
    _call("qspec_rms_norm_general_fuse_sum_fp16", _chk(out, "out", _F16), _chk(x, "x", _F16), float(eps), T, H,
          _stream())


def add_rms_norm_i4(out_q, scaling, hidden_out, x, delta, eps: float):
    """hidden_out = x + delta (fp16), then the i4 norm of hidden_out (quarot_llama.py:380-388 fused)."""
    H = x.shape[-1]
    T = x.numel() // H
    _call("qspec_add_rms_norm_i4", _chk(out_q, "out_q", _I8), _chk(scaling, "scaling", _F16),
          _opt(hidden_out, "hidden_out", _F16), _chk(x, "x", _F16), _opt(delta, "delta", _F16), float(eps), T, H,
          _stream())


def _xp_tile(t, name, K, tokens: int = 16):
    """A fragment-major activation tile always holds 16 rows (w4a16_act_layout_supported); 17..32 tokens: two tiles."""
    rows = 16 if tokens <= 16 else 32
    if t.numel() < rows * K:
        raise RuntimeError(f"{name}: {rows // 16} fragment-major tile(s) of 16 x {K} halves, got {tuple(t.shape)}")
    return t


def w4a16_act_layout_supported(M: int, K: int) -> bool:
    """True when the W4A16 GEMMs at (M <= 16 tokens, K) read the FRAGMENT-MAJOR activation tile the `xp=True` producers
    write (norm, head transform, MLP transform): [K/128][4][4 k-groups][16 rows][8 halves], the order the MFMA operand
    registers hold them, so the GEMM loads them straight from global memory with no LDS staging pass."""
    return bool(_lib.load().qspec_w4a16_act_layout_supported(M, K))


def w4a16_act_layout32_supported(M: int, N: int, K: int) -> bool:
    """17..32 tokens: the two-token-tile W4A16 streaming kernel takes (M, N, K); x is then two fragment-major tiles [2, 16, K]."""
    return bool(_lib.load().qspec_w4a16_act_layout32_supported(M, N, K))


def _xp32_tiles(t, name, K):
    if t.numel() < 32 * K:
        raise RuntimeError(f"{name}: two fragment-major tiles are 2 x 16 x {K} halves, got {tuple(t.shape)}")
    return _chk(t, name, _F16)


def w4a16_linear_xp32(x, wq, w_scale, out, tokens: int):
    N, K = wq.shape[0], wq.shape[1] * 2
    _call("qspec_w4a16_linear_xp32", _xp32_tiles(x, "x", K), _chk(wq, "wq", (_I8, _U8)), _chk(w_scale, "w_scale", _F16),
          _chk(out, "out", _F16), tokens, N, K, _stream())
    return out


def w4a16_linear_partial_slices_xp32(M: int, N: int, K: int) -> int:
    return int(_lib.load().qspec_w4a16_linear_partial_slices_xp32(M, N, K))


def w4a16_linear_partial_xp32(x, wq, part, slices: int, tokens: int):
    N, K = wq.shape[0], wq.shape[1] * 2
    _call("qspec_w4a16_linear_partial_xp32", _xp32_tiles(x, "x", K), _chk(wq, "wq", (_I8, _U8)), _chk(part, "part", _F32), tokens, N, K,
          slices, _stream())
    return part


def qkv_rope_linear_xp32(x, wq, w_scale, qkv, positions, cos_sin_cache, key_cache, value_cache, slot_mapping, num_heads,
                         num_kv_heads, head_size, tokens: int):
    N, K = wq.shape[0], wq.shape[1] * 2
    _call("qspec_qkv_rope_linear_w4a16_xp32", _xp32_tiles(x, "x", K), _chk(wq, "wq", (_I8, _U8)), _chk(w_scale, "w_scale", _F16),
          _chk(qkv, "qkv", _F16), tokens, N, K, _chk(positions, "positions", _I64), _chk(cos_sin_cache, "cos_sin_cache", _F16),
          _chk(key_cache, "key_cache", _F16), _chk(value_cache, "value_cache", _F16), _chk(slot_mapping, "slot_mapping", _I64),
          num_heads, num_kv_heads, head_size, cos_sin_cache.shape[-1], _stream())
    return qkv


def gate_up_silu_linear_xp32(x, wq, w_scale, act, tokens: int):
    I, K = wq.shape[0] // 2, wq.shape[1] * 2
    _call("qspec_gate_up_silu_linear_w4a16_xp32", _xp32_tiles(x, "x", K), _chk(wq, "wq", (_I8, _U8)),
          _chk(w_scale, "w_scale", _F16), _chk(act, "act", _F16), tokens, I, K, _stream())
    return act


def mlp_hadamard_act_layout_supported(T: int, I: int, K: int) -> bool:
    """mlp_hadamard(..., xp=True) exists at this shape (the spread forms and the one-workgroup form both store the tiles, so the
    answer does not depend on XWG_SPREAD: a recovery replay without hand-off kernels takes the same GEMM forms)."""
    return bool(_lib.load().qspec_mlp_hadamard_act_layout_supported(T, I, K))


def add_rms_norm_fp16(out, hidden_out, x, delta, eps: float, xp: bool = False):
    H = x.shape[-1]
    T = x.numel() // H
    if xp:
        _xp_tile(out, "out", H, T)
    _call("qspec_add_rms_norm_fp16" + ("_xp" if xp else ""), _chk(out, "out", _F16), _opt(hidden_out, "hidden_out", _F16),
          _chk(x, "x", _F16), _opt(delta, "delta", _F16), float(eps), T, H, _stream())


def fuse_sym_quant(x, scale, q, clip_ratio: float = 1.0):
    """quarot._CUDA.fuse_sym_quant(x, scale, q, clip) (third-party/QuaRot/quarot/kernels/bindings.cpp:128-147)."""
    T, K = x.shape
    _call("qspec_fuse_sym_quant", _chk(x, "x", _F16), _chk(scale, "scale", _F16), _chk(q, "q", _I8),
          float(clip_ratio), T, K, _stream())


# ------------------------------------------------------------------ hadamard

def faster_fast_hadamard_transform(x, scale: float, out):
    """fast_hadamard_transform_cuda.faster_fast_hadamard_transform(x, scale, out)
    (third-party/fast-hadamard-transform/csrc/fast_hadamard_transform.cpp:69-110)."""
    n = x.shape[-1]
    rows = x.numel() // n
    _call("qspec_fast_hadamard_transform", _chk(x, "x", _F16), float(scale), _chk(out, "out", _F16), rows, n,
          _stream())
    return out


def fast_hadamard_transform(x, scale: float = 1.0):
    """fast_hadamard_transform_cuda.fast_hadamard_transform(x, scale) -> new tensor (:113-154)."""
    return faster_fast_hadamard_transform(x, scale, torch.empty_like(x))


def hadamard_mix(y, hadK, out):
    """`hadK @ y.view(-1, K, m)` (quarot/functional/hadamard.py:104-108)."""
    T, K, m = y.shape
    _call("qspec_hadamard_mix", _chk(y, "y", _F16), _chk(hadK, "hadK", _F16), _chk(out, "out", _F16), T, K, m,
          _stream())
    return out


def heads_hadamard(attn, had_scale: float, out_f16=None, q=None, scale=None, clip_ratio: float = 1.0, heads=None):
    """quarot_llama.py:231-238 in one kernel; attn [T, heads, d] (or [T, heads*d] with heads given)."""
    if attn.dim() == 3:
        T, heads, d = attn.shape
    else:
        T = attn.shape[0]
        d = attn.shape[1] // heads
    _call("qspec_heads_hadamard", _chk(attn, "attn", _F16), _opt(out_f16, "out_f16", _F16), _opt(q, "q", _I8),
          _opt(scale, "scale", _F16), float(had_scale), float(clip_ratio), T, heads, d, _stream())


def heads_hadamard_mix(attn, hadK, K: int, had_scale: float, out):
    """Head transform for head counts K * 2^p with a table factor (matmul_hadU_cuda between the transposes of
    quarot_llama.py:231-234); attn [T, heads, d], out the same number of elements, fp16."""
    if attn.dim() != 3:
        raise RuntimeError("heads_hadamard_mix: attn must be [T, heads, d]")
    T, heads, d = attn.shape
    _call("qspec_heads_hadamard_mix", _chk(attn, "attn", _F16), _chk(hadK, "hadK", _F16), _chk(out, "out", _F16),
          float(had_scale), T, heads, d, K, _stream())
    return out


def silu_mul(gate_up, out):
    """`act_fn(gate) * up` over the fused gate_up row (quarot_llama.py:279-284)."""
    T, two_i = gate_up.shape
    _call("qspec_silu_mul", _chk(gate_up, "gate_up", _F16), _chk(out, "out", _F16), T, two_i // 2, _stream())
    return out


def silu_mul_hadamard(gate_up, hadK, K: int, had_scale: float, out_f16=None, q=None, scale=None,
                      clip_ratio: float = 1.0):
    """quarot_llama.py:279-295 in one kernel; gate_up [T, 2I] with up first."""
    T, two_i = gate_up.shape
    _call("qspec_silu_mul_hadamard", _chk(gate_up, "gate_up", _F16), _opt(hadK, "hadK", _F16),
          _opt(out_f16, "out_f16", _F16), _opt(q, "q", _I8), _opt(scale, "scale", _F16), float(had_scale),
          float(clip_ratio), T, two_i // 2, K, _stream())


_xwg_ws = {}
XWG_SPREAD = os.environ.get("QSPEC_XWG_SPREAD", "1") != "0"   # False: one workgroup per token (no workspace)


def xwg_workspace(device):
    """Exchange workspace of the kernels that spread one token over several workgroups (mlp_hadamard,
    heads_hadamard_merged): zero-filled once, never reset; one per (device, stream) -- launches on one stream are
    serialised, two streams must not share its counters.  Word 0 is a sticky error flag (xwg_error)."""
    if not XWG_SPREAD:
        return None
    key = _ws_key(device)
    if key not in _xwg_ws:
        _xwg_ws[key] = torch.zeros(int(_lib.load().qspec_xwg_workspace_bytes()) // 4, dtype=torch.int32, device=device)
    return _xwg_ws[key]


def xwg_error_word(device):
    """Sticky error flags of EVERY exchange workspace of the device (a captured graph keeps the workspace of its
    capture stream): an int32 tensor [n] (one word per workspace), or None if none exists."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    words = [ws[:1] for key, ws in _xwg_ws.items() if key[:2] == (device.type, idx)]
    return torch.cat(words) if words else None


def mlp_hadamard(act, hadK, K: int, had_scale: float, out_f16=None, q=None, scale=None, clip_ratio: float = 1.0,
                 workspace="auto", xp: bool = False):
    """Hadamard (+ quantiser) tail of silu_mul_hadamard on act = silu(gate)*up, [T, I]."""
    T, I = act.shape
    ws = xwg_workspace(act.device) if isinstance(workspace, str) else workspace
    if xp:
        _xp_tile(out_f16, "out_f16", I, T)
    _call("qspec_mlp_hadamard" + ("_xp" if xp else ""), _chk(act, "act", _F16), _opt(hadK, "hadK", _F16), _opt(out_f16, "out_f16", _F16),
          _opt(q, "q", _I8), _opt(scale, "scale", _F16), float(had_scale), float(clip_ratio), T, I, K,
          None if ws is None else ws.data_ptr(), _stream())


# ------------------------------------------------------------------ linear

def qkv_rope_linear_supported(w4a4: bool, M: int, N: int, K: int, head_size: int) -> bool:
    """The fused qkv GEMM + RoPE + KV write exists for head size 128 (every shape) and 64 (streaming shapes)."""
    return bool(_lib.load().qspec_qkv_rope_linear_supported(int(bool(w4a4)), M, N, K, head_size))


def qkv_rope_linear(x, x_scale, wq, w_scale, qkv, positions, cos_sin_cache, key_cache, value_cache, slot_mapping,
                    num_heads, num_kv_heads, head_size, xp: bool = False, tokens=None):
    """qkv GEMM + rotary_embedding + reshape_and_cache_flash in one launch (quarot_llama.py:183-226).
    x_scale is None -> W4A16 (x fp16 [M,K]); else W4A4 (x packed int4 [M,K/2])."""
    N = wq.shape[0]
    K = wq.shape[1] * 2
    M = x.shape[0] if tokens is None else tokens    # xp: x is the 16-row tile, tokens the rows in use
    if xp:
        _xp_tile(x, "x", K)
    common = (_chk(positions, "positions", _I64), _chk(cos_sin_cache, "cos_sin_cache", _F16),
              _chk(key_cache, "key_cache", _F16), _chk(value_cache, "value_cache", _F16),
              _chk(slot_mapping, "slot_mapping", _I64), num_heads, num_kv_heads, head_size, cos_sin_cache.shape[-1])
    if x_scale is not None:
        _call("qspec_qkv_rope_linear_s4s4", _chk(x, "xq", (_I8, _U8)), _chk(x_scale, "x_scale", _F16),
              _chk(wq, "wq", (_I8, _U8)), _chk(w_scale, "w_scale", _F16), _chk(qkv, "qkv", _F16), M, N, K, *common,
              _stream())
    else:
        _call("qspec_qkv_rope_linear_w4a16" + ("_xp" if xp else ""), _chk(x, "x", _F16), _chk(wq, "wq", (_I8, _U8)),
              _chk(w_scale, "w_scale", _F16), _chk(qkv, "qkv", _F16), M, N, K, *common,
              w4a16_workspace(x.device).data_ptr(), _stream())
    return qkv


def gate_up_silu_linear(x, x_scale, wq, w_scale, act, xp: bool = False, tokens=None):
    """gate_up GEMM + silu(gate)*up in one launch (quarot_llama.py:276-284); act [M, I]."""
    M = x.shape[0] if tokens is None else tokens
    I = wq.shape[0] // 2
    K = wq.shape[1] * 2
    if xp:
        _xp_tile(x, "x", K)
    if x_scale is not None:
        _call("qspec_gate_up_silu_linear_s4s4", _chk(x, "xq", (_I8, _U8)), _chk(x_scale, "x_scale", _F16),
              _chk(wq, "wq", (_I8, _U8)), _chk(w_scale, "w_scale", _F16), _chk(act, "act", _F16), M, I, K, _stream())
    else:
        _call("qspec_gate_up_silu_linear_w4a16" + ("_xp" if xp else ""), _chk(x, "x", _F16), _chk(wq, "wq", (_I8, _U8)),
              _chk(w_scale, "w_scale", _F16), _chk(act, "act", _F16), M, I, K, w4a16_workspace(x.device).data_ptr(),
              _stream())
    return act


def prefetch(t, workgroups: int = 128):
    """Cache hint (qspec_prefetch): pull tensor `t` into the Infinity Cache on the current stream.  EXPERIMENTAL build only
    (QSPEC_HIP_LIB=.../libqspec_hip_experimental.so): measured and left out of the engine."""
    _call("qspec_prefetch", t.data_ptr(), t.numel() * t.element_size(), workgroups, _stream())


def prefetch_tiles(wq, first_tile: int = 0, ntiles: Optional[int] = None, workgroups: int = 256):
    """Cache hint, tile aligned: workgroup b touches weight tiles b, b + workgroups, ... (16 rows each) of wq [N, K/2]."""
    tile_bytes = 16 * wq.shape[1] * wq.element_size()
    n = wq.shape[0] // 16 if ntiles is None else ntiles
    _call("qspec_prefetch_tiles", wq.data_ptr(), tile_bytes, first_tile, n, workgroups, _stream())


def ln_linear_s4s4_supported(M: int, N: int, K: int) -> bool:
    return bool(_lib.load().qspec_ln_linear_s4s4_supported(M, N, K))


_ln_ws = {}
LN_HANDOFF_MIN_M = int(os.environ.get("QSPEC_LN_HANDOFF_MIN_M", "8"))
LN_HANDOFF = os.environ.get("QSPEC_LN_HANDOFF", "1") != "0"   # False: every workgroup recomputes the norm (no workspace)


def ln_linear_workspace(device, M: int = 16):
    """Hand-off workspace of the LN-prologue GEMMs (zero-filled once; every call leaves it zeroed).  Measured on
    MI355X: up to 4 tokens every workgroup recomputing the norm is faster (10.4 vs 12.2 us for qkv_proj), from 8
    tokens the producer / hand-off form wins (15 vs 25 us at 16 tokens)."""
    if not LN_HANDOFF or M < LN_HANDOFF_MIN_M:
        return None
    key = _ws_key(device)
    if key not in _ln_ws:
        _ln_ws[key] = torch.zeros(int(_lib.load().qspec_ln_linear_workspace_bytes()), dtype=torch.uint8, device=device)
    return _ln_ws[key]


def ln_linear_error_word(device):
    """Sticky error words of every norm hand-off workspace of the device (int32 [n]) or None."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    words = [ws.view(torch.int32)[31:32] for key, ws in _ln_ws.items() if key[:2] == (device.type, idx)]
    return torch.cat(words) if words else None


def ln_qkv_rope_linear(hidden_in, delta, hidden_out, eps, wq, w_scale, qkv, positions, cos_sin_cache, key_cache,
                       value_cache, slot_mapping, num_heads, num_kv_heads, head_size):
    """Draft pass: hidden_out = hidden_in + delta; LN + int4 quant; qkv GEMM; RoPE; KV write -- one launch
    (quarot_llama.py:373-374 + 183-226)."""
    M, K = hidden_in.shape
    N = wq.shape[0]
    _call("qspec_ln_qkv_rope_linear_s4s4", _chk(hidden_in, "hidden_in", _F16), _opt(delta, "delta", _F16),
          _opt(hidden_out, "hidden_out", _F16), float(eps), _chk(wq, "wq", (_I8, _U8)), _chk(w_scale, "w_scale", _F16),
          _chk(qkv, "qkv", _F16), M, N, K, _chk(positions, "positions", _I64), _chk(cos_sin_cache, "cos_sin_cache", _F16),
          _chk(key_cache, "key_cache", _F16), _chk(value_cache, "value_cache", _F16),
          _chk(slot_mapping, "slot_mapping", _I64), num_heads, num_kv_heads, head_size, cos_sin_cache.shape[-1],
          _opt(ln_linear_workspace(hidden_in.device, M), "sync_workspace"), _stream())
    return qkv


def ln_gate_up_silu_linear(hidden_in, delta, hidden_out, eps, wq, w_scale, act):
    """Draft pass: hidden_out = hidden_in + delta; LN + int4 quant; gate_up GEMM; silu(gate)*up -- one launch
    (quarot_llama.py:380-388 + 276-284)."""
    M, K = hidden_in.shape
    I = wq.shape[0] // 2
    _call("qspec_ln_gate_up_silu_linear_s4s4", _chk(hidden_in, "hidden_in", _F16), _opt(delta, "delta", _F16),
          _opt(hidden_out, "hidden_out", _F16), float(eps), _chk(wq, "wq", (_I8, _U8)), _chk(w_scale, "w_scale", _F16),
          _chk(act, "act", _F16), M, I, K, _opt(ln_linear_workspace(hidden_in.device, M), "sync_workspace"), _stream())
    return act


def rowwise_scaled_linear_cutlass_s4s4_unified(xq, x_scale, wq, w_scale, bias, out):
    """torch.ops.torchao.rowwise_scaled_linear_cutlass_s4s4_unified (third-party/ao/torchao/ops.py:600-636)."""
    M, Kb = xq.shape
    N = wq.shape[0]
    if wq.shape[1] != Kb:
        raise RuntimeError(f"xq and wq disagree on K: {xq.shape} vs {wq.shape}")
    if out.shape[0] != M or out.shape[1] != N:
        raise RuntimeError(f"out has shape {tuple(out.shape)}, expected ({M}, {N})")
    _call("qspec_rowwise_scaled_linear_s4s4", _chk(xq, "xq", (_I8, _U8)), _chk(x_scale, "x_scale", _F16),
          _chk(wq, "wq", (_I8, _U8)), _chk(w_scale, "w_scale", _F16), _opt(bias, "bias", _F16),
          _chk(out, "out", _F16), M, N, Kb * 2, _stream())
    return out


def rowwise_scaled_linear_s4s4_residual(xq, x_scale, wq, w_scale, resid_in, resid_out):
    """resid_out = resid_in + linear_s4s4(xq) (fp16 add of the fp16 GEMM result) in one launch (quarot_llama.py:380,390)."""
    M, Kb = xq.shape
    N = wq.shape[0]
    _call("qspec_rowwise_scaled_linear_s4s4_residual", _chk(xq, "xq", (_I8, _U8)), _chk(x_scale, "x_scale", _F16),
          _chk(wq, "wq", (_I8, _U8)), _chk(w_scale, "w_scale", _F16), _chk(resid_in, "resid_in", _F16),
          _chk(resid_out, "resid_out", _F16), M, N, Kb * 2, _stream())
    return resid_out


def rowwise_scaled_linear_s4s4_residual_supported(M: int, N: int, K: int) -> bool:
    return bool(_lib.load().qspec_rowwise_scaled_linear_s4s4_residual_supported(M, N, K))


def rowwise_scaled_linear_s4s4_residual_hq(x16, part_amax, clip_ratio, wq, w_scale, resid_in, resid_out):
    """rowwise_scaled_linear_s4s4_residual on UNQUANTISED fp16 rows: the row-absmax int4 quantiser (quarot_llama.py:235-238,
    quant.cu:102-167) runs in the GEMM's prologue from `part_amax` [M, n_parts] partial row maxima."""
    M, K = x16.shape
    N = wq.shape[0]
    _call("qspec_rowwise_scaled_linear_s4s4_residual_hq", _chk(x16, "x16", _F16), _chk(part_amax, "part_amax", _F32),
          part_amax.shape[1], float(clip_ratio), _chk(wq, "wq", (_I8, _U8)), _chk(w_scale, "w_scale", _F16),
          _chk(resid_in, "resid_in", _F16), _chk(resid_out, "resid_out", _F16), M, N, K, _stream())
    return resid_out


def rowwise_scaled_linear_s4s4_residual_hq_supported(M: int, N: int, K: int, n_parts: int = 8) -> bool:
    return bool(_lib.load().qspec_rowwise_scaled_linear_s4s4_residual_hq_supported(M, N, K, n_parts))


_w16_ws = {}


def w4a16_workspace(device):
    """Zero-initialised split-K scratch of the W4A16 kernels, one per (device, stream): calls on one stream are
    serialised, so they can share it (every call leaves the ticket counters at zero); two streams must not."""
    key = _ws_key(device)
    if key not in _w16_ws:
        _w16_ws[key] = torch.zeros(int(_lib.load().qspec_w4a16_workspace_bytes()), dtype=torch.uint8, device=device)
    return _w16_ws[key]


def w4a16_linear(x, wq, w_scale, out, bias=None, xp: bool = False, tokens=None):
    """bitblas.Matmul(x, w ^ 0x88, output=out, scale=w_scale, bias=bias) on the SAME packed buffer
    (quarot_nn/linear.py:122)."""
    M, K = x.shape
    N = wq.shape[0]
    if wq.shape[1] * 2 != K:
        raise RuntimeError(f"x and wq disagree on K: {x.shape} vs {wq.shape}")
    if xp:
        M = M if tokens is None else tokens
        _xp_tile(x, "x", K)
    _call("qspec_w4a16_linear" + ("_xp" if xp else ""), _chk(x, "x", _F16), _chk(wq, "wq", (_I8, _U8)), _chk(w_scale, "w_scale", _F16),
          _opt(bias, "bias", _F16), _chk(out, "out", _F16), M, N, K, w4a16_workspace(x.device).data_ptr(), _stream())
    return out


def w4a16_linear_partial_slices(M: int, N: int, K: int) -> int:
    """Slices the long-K W4A16 path cuts K into for this shape (0: use w4a16_linear)."""
    return int(_lib.load().qspec_w4a16_linear_partial_slices(M, N, K))


def w4a16_linear_partial(x, wq, part, slices: int, xp: bool = False, tokens=None):
    """Raw fp32 K-slice sums of x @ dequant(wq)^T into part [slices, M, N] (finished by add_rms_norm_fp16_partial)."""
    M, K = x.shape
    N = wq.shape[0]
    if xp:
        M = M if tokens is None else tokens
        _xp_tile(x, "x", K)
    _call("qspec_w4a16_linear_partial" + ("_xp" if xp else ""), _chk(x, "x", _F16), _chk(wq, "wq", (_I8, _U8)), _chk(part, "part", _F32), M, N, K,
          slices, _stream())
    return part


def add_rms_norm_fp16_partial(out, hidden_out, x, part, w_scale, slices: int, eps: float, xp: bool = False):
    """hidden_out = x + h(sum_s part[s] * w_scale); out = LN(hidden_out) -- the finish of w4a16_linear_partial fused in."""
    H = x.shape[-1]
    T = x.numel() // H
    if xp:
        _xp_tile(out, "out", H, T)
    _call("qspec_add_rms_norm_fp16_partial" + ("_xp" if xp else ""), _chk(out, "out", _F16), _chk(hidden_out, "hidden_out", _F16),
          _chk(x, "x", _F16), _chk(part, "part", _F32), _chk(w_scale, "w_scale", _F16), slices, float(eps), T, H, _stream())


def rowwise_scaled_linear_s4s4_partial_slices(M: int, N: int, K: int) -> int:
    """K slices the s4s4 linear is cut into at this shape (17..32 tokens, one weight tile per workgroup); 0: use the plain entry."""
    return int(_lib.load().qspec_rowwise_scaled_linear_s4s4_partial_slices(M, N, K))


def rowwise_scaled_linear_s4s4_partial(xq, wq, ipart, slices: int):
    """Raw int32 K-slice sums of xq @ wq^T into ipart [slices, M, N] (finished by add_rms_norm_ipartial)."""
    M, N, K = xq.shape[0], wq.shape[0], 2 * xq.shape[1]
    _call("qspec_rowwise_scaled_linear_s4s4_partial", _chk(xq, "xq", (_I8, _U8)), _chk(wq, "wq", (_I8, _U8)),
          _chk(ipart, "ipart", _I32), M, N, K, slices, _stream())
    return ipart


def add_rms_norm_ipartial(hidden_out, x, ipart, x_scale, w_scale, slices: int, eps: float, q=None, scale=None, out_f16=None):
    """hidden_out = x + h((sum_s ipart[s]) * x_scale[row] * w_scale[col]) (the s4s4 epilogue + residual add); then the norm of
    hidden_out: int4 rows q + scale, or fp16 rows out_f16."""
    H = x.shape[-1]
    T = x.numel() // H
    _call("qspec_add_rms_norm_ipartial", _opt(q, "q", _I8), _opt(scale, "scale", _F16), _opt(out_f16, "out_f16", _F16),
          _chk(hidden_out, "hidden_out", _F16), _chk(x, "x", _F16), _chk(ipart, "ipart", _I32), _chk(x_scale, "x_scale", _F16),
          _chk(w_scale, "w_scale", _F16), slices, float(eps), T, H, _stream())


def w4a16_linear_ksliced(x, wq, w_scale, out, k0: int, k1: int):
    """Row-parallel shard: out = x[:, k0:k1] @ dequant(wq)[:, k0:k1]^T * w_scale (partial sum; caller all-reduces).
    x [M,K] and wq [N,K/2] are the FULL tensors; only the K range is read."""
    M, K = x.shape
    N = wq.shape[0]
    _chk(x, "x", _F16); _chk(wq, "wq", (_I8, _U8))
    _call("qspec_w4a16_linear_ksliced", x.data_ptr() + 2 * k0, K, wq.data_ptr() + k0 // 2, K // 2,
          _chk(w_scale, "w_scale", _F16), _chk(out, "out", _F16), M, N, k1 - k0, w4a16_workspace(x.device).data_ptr(),
          _stream())
    return out


def w4a16_linear_ksliced_raw(x, wq, part, k0: int, k1: int):
    """Row-parallel shard as raw fp32 sums: part [M, N] = x[:, k0:k1] @ dequant(wq)[:, k0:k1]^T (no scale, no
    rounding); reduced across ranks in fp32, finished by add_rms_norm_fp16_partial(..., slices=1)."""
    M, K = x.shape
    N = wq.shape[0]
    _chk(x, "x", _F16); _chk(wq, "wq", (_I8, _U8))
    _call("qspec_w4a16_linear_ksliced_raw", x.data_ptr() + 2 * k0, K, wq.data_ptr() + k0 // 2, K // 2,
          _chk(part, "part", _F32), M, N, k1 - k0, w4a16_workspace(x.device).data_ptr(), _stream())
    return part


def gate_up_silu_linear_shard(x, wq, w_scale, act, ch0: int, nch: int):
    """Column-parallel shard of the fused gate_up + silu*up: writes act[:, ch0:ch0+nch] (act is the full [M, I])."""
    M = x.shape[0]
    I = wq.shape[0] // 2
    K = wq.shape[1] * 2
    _call("qspec_gate_up_silu_linear_w4a16_shard", _chk(x, "x", _F16), _chk(wq, "wq", (_I8, _U8)),
          _chk(w_scale, "w_scale", _F16), _chk(act, "act", _F16), M, I, K, ch0, nch,
          w4a16_workspace(x.device).data_ptr(), _stream())
    return act


def linear_f16(x, w, out):
    M, K = x.shape
    N = w.shape[0]
    _call("qspec_linear_f16", _chk(x, "x", _F16), _chk(w, "w", _F16), _chk(out, "out", _F16), M, N, K, _stream())
    return out


def dequant_w4(wq, w_scale, out):
    N, Kb = wq.shape
    _call("qspec_dequant_w4", _chk(wq, "wq", (_I8, _U8)), _chk(w_scale, "w_scale", _F16), _chk(out, "out", _F16), N,
          Kb * 2, _stream())
    return out


# ------------------------------------------------------------------ attention side

def rotary_embedding(positions, query, key, head_size: int, cos_sin_cache, is_neox: bool = True):
    """torch.ops._C.rotary_embedding (csrc/pos_encoding_kernels.cu:124); query/key [T, n*head_size], last dim contiguous."""
    if not is_neox:
        raise RuntimeError("only the NeoX layout is on the QSpec path (quarot_llama.py:104)")
    T = positions.numel()
    for t, n in ((query, "query"), (key, "key")):
        if t.stride(-1) != 1 or t.dtype != _F16 or not t.is_cuda:
            raise RuntimeError(f"{n}: need fp16 HIP tensor with unit inner stride")
    nq, nk = query.shape[-1] // head_size, key.shape[-1] // head_size
    _call("qspec_rotary_embedding", _chk(positions, "positions", _I64), query.data_ptr(), key.data_ptr(),
          _chk(cos_sin_cache, "cos_sin_cache", _F16), T, nq, nk, head_size, cos_sin_cache.shape[-1],
          query.stride(0), key.stride(0), _stream())


def reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping):
    """torch.ops._C_cache_ops.reshape_and_cache_flash (csrc/cache_kernels.cu:304); key/value [T, n_kv, d]."""
    T, nkv, d = key.shape
    _call("qspec_reshape_and_cache_flash", key.data_ptr(), value.data_ptr(), _chk(key_cache, "key_cache", _F16),
          _chk(value_cache, "value_cache", _F16), _chk(slot_mapping, "slot_mapping", _I64), T, nkv, d, key.stride(0),
          value.stride(0), _stream())


def rope_kv_write(positions, qkv, cos_sin_cache, key_cache, value_cache, slot_mapping, num_heads, num_kv_heads,
                  head_size):
    T = qkv.shape[0]
    _call("qspec_rope_kv_write", _chk(positions, "positions", _I64), _chk(qkv, "qkv", _F16),
          _chk(cos_sin_cache, "cos_sin_cache", _F16), _chk(key_cache, "key_cache", _F16),
          _chk(value_cache, "value_cache", _F16), _chk(slot_mapping, "slot_mapping", _I64), T, num_heads,
          num_kv_heads, head_size, cos_sin_cache.shape[-1], _stream())


def paged_attention_workspace_bytes(max_tokens, num_heads, head_size, n_splits) -> int:
    return int(_lib.load().qspec_paged_attention_workspace_bytes(max_tokens, num_heads, head_size, n_splits))


def paged_attention(q, q_stride, key_cache, value_cache, block_tables, ctx_lens, q_start, tokens, max_q_len,
                    num_heads, sm_scale, n_splits, workspace, out):
    """flash_attn_with_kvcache / flash_attn_varlen_func over the paged cache (flash_attn.py:741-830)."""
    nb, bs, nkv, d = key_cache.shape
    n_seqs = ctx_lens.numel()
    _call("qspec_paged_attention", q.data_ptr(), q_stride, _chk(key_cache, "key_cache", _F16),
          _chk(value_cache, "value_cache", _F16), _chk(block_tables, "block_tables", _I32), block_tables.shape[1],
          _chk(ctx_lens, "ctx_lens", _I32), _chk(q_start, "q_start", _I32), n_seqs, tokens, max_q_len, num_heads, nkv,
          d, bs, float(sm_scale), n_splits, workspace.data_ptr(), _opt(out, "out", _F16), _stream())


def heads_hadamard_merged(workspace, max_tokens, n_splits, tokens, heads, head_dim, had_scale: float, out_f16=None,
                          q=None, scale=None, clip_ratio: float = 1.0, xp: bool = False):
    """Split merge of paged_attention(..., out=None) + heads_hadamard in one launch."""
    if xp:
        _xp_tile(out_f16, "out_f16", heads * head_dim, tokens)
    _call("qspec_heads_hadamard_merged" + ("_xp" if xp else ""), workspace.data_ptr(), max_tokens, n_splits, _opt(out_f16, "out_f16", _F16),
          _opt(q, "q", _I8), _opt(scale, "scale", _F16), float(had_scale), float(clip_ratio), tokens, heads, head_dim,
          _stream())


def heads_hadamard_merged_spread(workspace, max_tokens, n_splits, tokens, heads, head_dim, had_scale: float, out_f16, part_amax):
    """heads_hadamard_merged spread over 8 workgroups per token: fp16 rows + part_amax [tokens, 8] (the quantiser then runs in
    rowwise_scaled_linear_s4s4_residual_hq)."""
    _call("qspec_heads_hadamard_merged_spread", workspace.data_ptr(), max_tokens, n_splits, _chk(out_f16, "out_f16", _F16),
          _chk(part_amax, "part_amax", _F32), float(had_scale), tokens, heads, head_dim, _stream())


def heads_hadamard_merged_spread_supported(tokens: int, heads: int, head_dim: int) -> bool:
    return bool(_lib.load().qspec_heads_hadamard_merged_spread_supported(tokens, heads, head_dim))


def heads_hadamard_mix_merged_spread(workspace, max_tokens, n_splits, tokens, heads, head_dim, hadK, K: int, had_scale: float,
                                     out_f16, part_amax=None, xp: bool = False):
    """The spread merge + head transform for head counts with a table factor (40 heads = had40): fp16 rows, and with
    part_amax [tokens, 8] the partial row maxima for rowwise_scaled_linear_s4s4_residual_hq."""
    if xp:
        _xp_tile(out_f16, "out_f16", heads * head_dim, tokens)
    _call("qspec_heads_hadamard_mix_merged_spread" + ("_xp" if xp else ""), workspace.data_ptr(), max_tokens, n_splits, _chk(hadK, "hadK", _F16), K,
          _chk(out_f16, "out_f16", _F16), _opt(part_amax, "part_amax", _F32), float(had_scale), tokens, heads, head_dim,
          _stream())


def heads_hadamard_mix_merged_spread_supported(tokens: int, heads: int, head_dim: int, K: int) -> bool:
    return bool(_lib.load().qspec_heads_hadamard_mix_merged_spread_supported(tokens, heads, head_dim, K))


# ------------------------------------------------------------------ token side

def embedding(ids, table, out):
    T = ids.numel()
    V, H = table.shape
    _call("qspec_embedding", _chk(ids, "ids", _I64), _chk(table, "table", _F16), _chk(out, "out", _F16), T, H, V,
          _stream())
    return out


_ws_cache = {}


def _sampler_ws(rows: int, device, workspace=None):
    """Scratch for the chunked softmax / rejection kernels (64 partials per row); cached per (device, stream, rows)."""
    if workspace is not None:
        return workspace
    key = (_ws_key(device), rows)
    if key not in _ws_cache:
        n = int(_lib.load().qspec_sampler_workspace_bytes(rows))
        _ws_cache[key] = torch.empty(n, dtype=torch.uint8, device=device)
    return _ws_cache[key]


def softmax_argmax(logits, probs, token, workspace=None):
    T, V = logits.shape
    ws = _sampler_ws(T, logits.device, workspace)
    _call("qspec_softmax_argmax", _chk(logits, "logits", _F16), _chk(probs, "probs", _F32), _chk(token, "token", _I64),
          T, V, ws.data_ptr(), _stream())


_head_ws = {}


def lm_head_softmax_argmax_supported(T: int, V: int, K: int) -> bool:
    return bool(_lib.load().qspec_lm_head_softmax_argmax_supported(T, V, K))


def lm_head_softmax_argmax(hidden, lm_head, logits, probs, token):
    """logits = hidden @ lm_head^T (fp16, scratch), probs = softmax(float(logits)), token = argmax -- the lm_head launch
    hands its row maxima to the softmax, probs is written once (logits_processor.py:92-97 + sampler.py:270-287)."""
    T, K = hidden.shape
    V = lm_head.shape[0]
    key = (_ws_key(hidden.device), T)
    if key not in _head_ws:
        _head_ws[key] = torch.empty(int(_lib.load().qspec_lm_head_sampler_workspace_bytes(T)), dtype=torch.uint8,
                                    device=hidden.device)
    _call("qspec_lm_head_softmax_argmax", _chk(hidden, "hidden", _F16), _chk(lm_head, "lm_head", _F16),
          _chk(logits, "logits", _F16), _chk(probs, "probs", _F32), _chk(token, "token", _I64), T, V, K,
          _head_ws[key].data_ptr(), _stream())


def rejection_sample(target_with_bonus_probs, bonus_token_ids, draft_probs, draft_token_ids, out_tokens, accepted,
                     recovered, counters=None, uniform=None, exponential=None, seed: int = 0, offset: int = 0,
                     rng_state=None, workspace=None, active_lens=None):
    B, k, V = draft_probs.shape
    # draft_probs / draft_token_ids / bonus_token_ids may be strided views (step-major draft buffers)
    if draft_probs.stride(2) != 1 or draft_probs.dtype != _F32 or draft_token_ids.dtype != _I64 \
            or bonus_token_ids.dtype != _I64 or not draft_probs.is_cuda:
        raise RuntimeError("draft_probs must be fp32 with unit inner stride; ids int64; all on the GPU")
    _call("qspec_rejection_sample", _chk(target_with_bonus_probs, "target_with_bonus_probs", _F32),
          bonus_token_ids.data_ptr(), draft_probs.data_ptr(), draft_token_ids.data_ptr(),
          _opt(uniform, "uniform", _F32), _opt(exponential, "exponential", _F32), seed, offset,
          _opt(rng_state, "rng_state", _I64), B, k, V, draft_probs.stride(0), draft_probs.stride(1),
          draft_token_ids.stride(0), draft_token_ids.stride(1),
          bonus_token_ids.stride(0) if bonus_token_ids.numel() > 1 else 1, _chk(out_tokens, "out_tokens", _I64),
          _chk(accepted, "accepted", _U8), _chk(recovered, "recovered", _I64), _opt(counters, "counters", _I64),
          _opt(active_lens, "active_lens", _I32), _sampler_ws(B * k, draft_probs.device, workspace).data_ptr(), _stream())


_sample_ws = {}


def sample_workspace(rows: int, device):
    """Histogram workspace of sample_top_k_top_p: zero-filled once per (device, stream, rows), left zeroed by every call."""
    key = _ws_key(device) + (rows,)
    if key not in _sample_ws:
        _sample_ws[key] = torch.zeros(int(_lib.load().qspec_sample_workspace_bytes(rows)), dtype=torch.uint8, device=device)
    return _sample_ws[key]


def sample_top_k_top_p(logits, probs, token, temperature=None, top_k=None, top_p=None, exponential=None, seed: int = 0,
                       offset: int = 0, rng_state=None, workspace=None):
    """Sampler.forward for non-greedy rows (sampler.py:216-316): temperature, top-k / top-p masking, softmax, multinomial by
    exponential noise.  temperature / top_p [T] fp32, top_k [T] int32 on the device (None: 1.0 / off)."""
    T, V = logits.shape
    ws = workspace if workspace is not None else sample_workspace(T, logits.device)
    _call("qspec_sample_top_k_top_p", _chk(logits, "logits", _F16), _opt(temperature, "temperature", _F32),
          _opt(top_k, "top_k", _I32), _opt(top_p, "top_p", _F32), _opt(exponential, "exponential", _F32), seed, offset,
          _opt(rng_state, "rng_state", _I64), _chk(probs, "probs", _F32), token.data_ptr(), token.stride(0) if token.dim() else 1,
          T, V, ws.data_ptr(), _stream())


def typical_acceptance_sample(target_with_bonus_probs, bonus_token_ids, draft_token_ids, posterior_threshold: float,
                              posterior_alpha: float, out_tokens, accepted, recovered, counters=None, workspace=None,
                              active_lens=None):
    B, k1, V = target_with_bonus_probs.shape
    k = k1 - 1
    if draft_token_ids.dtype != _I64 or bonus_token_ids.dtype != _I64 or not target_with_bonus_probs.is_cuda:
        raise RuntimeError("ids must be int64; all tensors on the GPU")
    _call("qspec_typical_acceptance_sample", _chk(target_with_bonus_probs, "target_with_bonus_probs", _F32),
          bonus_token_ids.data_ptr(), draft_token_ids.data_ptr(), float(posterior_threshold), float(posterior_alpha), B, k, V,
          draft_token_ids.stride(0), draft_token_ids.stride(1),
          bonus_token_ids.stride(0) if bonus_token_ids.numel() > 1 else 1, _chk(out_tokens, "out_tokens", _I64),
          _chk(accepted, "accepted", _U8), _chk(recovered, "recovered", _I64), _opt(counters, "counters", _I64),
          _opt(active_lens, "active_lens", _I32), _sampler_ws(B * k, target_with_bonus_probs.device, workspace).data_ptr(), _stream())


def advance_step_flashattn(num_seqs, num_queries, block_size, input_tokens, sampled_token_ids, input_positions,
                           seq_lens, slot_mapping, block_tables):
    """ops.advance_step_flashattn (vllm/_custom_ops.py; csrc/prepare_inputs/advance_step.cu:192)."""
    if num_seqs != num_queries:
        raise RuntimeError("CUDA-graph padding (num_seqs != num_queries) is not used on this path")
    _call("qspec_advance_step_flashattn", num_seqs, block_size, _chk(input_tokens, "input_tokens", _I64),
          _chk(sampled_token_ids, "sampled_token_ids", _I64), _chk(input_positions, "input_positions", _I64),
          _chk(seq_lens, "seq_lens", _I32), _chk(slot_mapping, "slot_mapping", _I64),
          _chk(block_tables, "block_tables", _I32), block_tables.stride(0), _stream())


# ------------------------------------------------------------------ spec-decode cycle glue

def _embed_args(embed):
    """embed = (embed_tokens [V, H], hidden_out [rows, H]): the forward's embedding lookup rides in the bookkeeping launch."""
    table, out = embed
    return (_chk(table, "embed_tokens", _F16), _chk(out, "hidden_out", _F16), table.shape[1], table.shape[0])


def spec_advance_draft(block_size, input_tokens, sampled_token_ids, positions, ctx_lens, slot_mapping, block_tables, embed=None):
    """_gpu_advance_step between two draft steps (draft_model_runner.py:78-135) with the engine's empty-slot and
    out-of-blocks rules (include/qspec_hip.h)."""
    B = ctx_lens.numel()
    if embed is not None:
        _call("qspec_spec_advance_draft_embed", B, block_size, block_tables.shape[1], _chk(input_tokens, "input_tokens", _I64),
              _chk(sampled_token_ids, "sampled_token_ids", _I64), _chk(positions, "positions", _I64),
              _chk(ctx_lens, "ctx_lens", _I32), _chk(slot_mapping, "slot_mapping", _I64),
              _chk(block_tables, "block_tables", _I32), block_tables.stride(0), *_embed_args(embed), _stream())
        return
    _call("qspec_spec_advance_draft", B, block_size, block_tables.shape[1], _chk(input_tokens, "input_tokens", _I64),
          _chk(sampled_token_ids, "sampled_token_ids", _I64), _chk(positions, "positions", _I64),
          _chk(ctx_lens, "ctx_lens", _I32), _chk(slot_mapping, "slot_mapping", _I64),
          _chk(block_tables, "block_tables", _I32), block_tables.stride(0), _stream())


def spec_prepare_draft(last_token, seq_lens, block_tables, block_size, input_tokens, positions, slot_mapping,
                       ctx_lens, embed=None, step_mask=None, eff_lens=None):
    """embed given: the fused form; step_mask / eff_lens (fused form only): eff_lens = seq_lens * step_mask, used as the lengths."""
    B = seq_lens.numel()
    if embed is not None:
        _call("qspec_spec_prepare_draft_embed", B, block_size, block_tables.shape[1], _chk(last_token, "last_token", _I64),
              _chk(seq_lens, "seq_lens", _I32), _opt(step_mask, "step_mask", _I32), _opt(eff_lens, "eff_lens", _I32),
              _chk(block_tables, "block_tables", _I32), block_tables.stride(0),
              _chk(input_tokens, "input_tokens", _I64), _chk(positions, "positions", _I64),
              _chk(slot_mapping, "slot_mapping", _I64), _chk(ctx_lens, "ctx_lens", _I32), *_embed_args(embed), _stream())
        return
    if step_mask is not None:
        raise RuntimeError("step_mask rides in the fused form only (embed=...)")
    _call("qspec_spec_prepare_draft", B, block_size, block_tables.shape[1], _chk(last_token, "last_token", _I64),
          _chk(seq_lens, "seq_lens", _I32), _chk(block_tables, "block_tables", _I32), block_tables.stride(0),
          _chk(input_tokens, "input_tokens", _I64), _chk(positions, "positions", _I64),
          _chk(slot_mapping, "slot_mapping", _I64), _chk(ctx_lens, "ctx_lens", _I32), _stream())


def spec_prepare_verify(last_token, draft_token_ids, seq_lens, block_tables, block_size, tokens, positions,
                        slot_mapping, ctx_lens, embed=None):
    B, k = draft_token_ids.shape
    if embed is not None:
        _call("qspec_spec_prepare_verify_embed", B, k, block_size, block_tables.shape[1], _chk(last_token, "last_token", _I64),
              draft_token_ids.data_ptr(), draft_token_ids.stride(0), draft_token_ids.stride(1),
              _chk(seq_lens, "seq_lens", _I32),
              _chk(block_tables, "block_tables", _I32), block_tables.stride(0), _chk(tokens, "tokens", _I64),
              _chk(positions, "positions", _I64), _chk(slot_mapping, "slot_mapping", _I64),
              _chk(ctx_lens, "ctx_lens", _I32), *_embed_args(embed), _stream())
        return
    _call("qspec_spec_prepare_verify", B, k, block_size, block_tables.shape[1], _chk(last_token, "last_token", _I64),
          draft_token_ids.data_ptr(), draft_token_ids.stride(0), draft_token_ids.stride(1),
          _chk(seq_lens, "seq_lens", _I32),
          _chk(block_tables, "block_tables", _I32), block_tables.stride(0), _chk(tokens, "tokens", _I64),
          _chk(positions, "positions", _I64), _chk(slot_mapping, "slot_mapping", _I64),
          _chk(ctx_lens, "ctx_lens", _I32), _stream())


def spec_commit(out_tokens, seq_lens, last_token, gen_tokens=None, gen_lens=None):
    B, k1 = out_tokens.shape
    cap = gen_tokens.shape[1] if gen_tokens is not None else 0
    _call("qspec_spec_commit", B, k1 - 1, _chk(out_tokens, "out_tokens", _I64), _chk(seq_lens, "seq_lens", _I32),
          _chk(last_token, "last_token", _I64), _opt(gen_tokens, "gen_tokens", _I64),
          _opt(gen_lens, "gen_lens", _I32), cap, _stream())


def spec_snapshot(seq_lens, gen_lens, last_token, counters, rng_state, snap_i32, snap_i64, restore: bool = False):
    """The small per-cycle sequence state copied aside (or back): what a recovery replay of a cycle starts from."""
    B = seq_lens.numel()
    assert snap_i32.numel() >= 2 * B and snap_i64.numel() >= B + 5 and counters.numel() == 3 and rng_state.numel() == 2
    _call("qspec_spec_snapshot", B, 1 if restore else 0, _chk(seq_lens, "seq_lens", _I32), _chk(gen_lens, "gen_lens", _I32),
          _chk(last_token, "last_token", _I64), _chk(counters, "counters", _I64), _chk(rng_state, "rng_state", _I64),
          _chk(snap_i32, "snap_i32", _I32), _chk(snap_i64, "snap_i64", _I64), _stream())


def stream_error_words(device, extra=()):
    """Raw device addresses of the sticky error words of the CURRENT stream's hand-off workspaces (the ones the launches
    being enqueued on this stream use; a captured graph keeps those of its capture stream) plus `extra` addresses (the
    one-shot all-reduce's word)."""
    key = _ws_key(device)
    ptrs = []
    if key in _xwg_ws:
        ptrs.append(_xwg_ws[key][:1].data_ptr())
    if key in _ln_ws:
        ptrs.append(_ln_ws[key].view(torch.int32)[31:32].data_ptr())
    ptrs += [int(p) for p in extra if p]
    return ptrs


def collect_error_words(word_addrs, out=None, clear: bool = False):
    """out[0] (int64) = OR of the sticky error words at `word_addrs` (stream_error_words; at most four); clear: zero them."""
    assert len(word_addrs) <= 4, "at most four error words per call"
    w = list(word_addrs) + [None] * (4 - len(word_addrs))
    _call("qspec_collect_error_words", w[0], w[1], w[2], w[3], 1 if clear else 0, _opt(out, "out", _I64), _stream())
