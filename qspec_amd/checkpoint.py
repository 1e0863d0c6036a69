"""Loader / exporter for the reference's on-disk QSpec checkpoint format.

The reference reads two safetensors shards, renames three key fragments and fuses the projections after the load
(vllm/worker/model_runner.py:1132-1148; fuse_qkv vllm/model_executor/models/quarot_llama.py:152-173, fuse_gate_up
:301-314).  Tensors per decoder layer `model.layers.{i}.`:

    self_attn.{q,k,v}_proj.weight          uint8 [N, K/2]   two's-complement int4 pairs, low nibble = even k
    self_attn.{q,k,v}_proj.weight_scales   fp16  [N, 1]
    self_attn.o_proj.1.weight / .weight_scales              (`o_proj.1.` -> `o_proj.` on load)
    mlp.{up,gate}_proj.weight / .weight_scales
    mlp.down_proj.0.had_rem_dim            fp16/fp32 [28, 28]  (`down_proj.0.` -> `online_hadamard.`)
    mlp.down_proj.2.weight / .weight_scales                    (`down_proj.2.` -> `down_proj.`)
  plus `model.embed_tokens.weight` and `lm_head.weight` (fp16); no norm weights (folded in offline,
  third-party/QuaRot/e2e/checkpoint_utils/quantize_llama_checkpoint.py:92-99).

`load_qspec_checkpoint` fills a qspec_amd.model.QuarotLlamaForCausalLM in place: q/k/v rows concatenated into
qkv_proj ([q; k; v]), up/gate into gate_up ([up; gate]) -- the fused layout every kernel of the hot path reads.
`save_qspec_checkpoint` writes the same format back (used by the tests; no reference checkpoint is reachable offline).
Only `safetensors` is used to touch the files (nothing in them is executed).
"""
from __future__ import annotations

import glob
import os
from typing import Dict, Iterable

import torch

RENAMES = (("o_proj.1.", "o_proj."), ("down_proj.0.", "online_hadamard."), ("down_proj.2.", "down_proj."))


def _rename(key: str) -> str:
    for a, b in RENAMES:        # model_runner.py:1138
        key = key.replace(a, b)
    return key


def read_state_dict(paths: Iterable[str]) -> Dict[str, torch.Tensor]:
    from safetensors.torch import load_file
    sd: Dict[str, torch.Tensor] = {}
    for p in paths:
        sd.update(load_file(p))
    return {_rename(k): v for k, v in sd.items()}


def checkpoint_files(model_dir: str):
    files = sorted(glob.glob(os.path.join(model_dir, "model-*-of-*.safetensors"))) or sorted(
        glob.glob(os.path.join(model_dir, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no safetensors shards under {model_dir}")
    return files


@torch.no_grad()
def load_state_dict(model, sd: Dict[str, torch.Tensor], strict: bool = True):
    """Fill `model` (QuarotLlamaForCausalLM) from a reference-layout state dict (keys already renamed)."""
    cfg = model.config
    used = set()

    def take(key, shape=None):
        if key not in sd:
            raise KeyError(f"checkpoint has no tensor `{key}`")
        t = sd[key]
        used.add(key)
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"`{key}` has shape {tuple(t.shape)}, expected {tuple(shape)}")
        return t

    def put_linear(lin, parts, prefix):
        """parts: reference sub-projection names concatenated along N in this order (fuse_qkv / fuse_gate_up)."""
        row = 0
        kb = lin.weight.shape[1]
        for name in parts:
            w = take(f"{prefix}.{name}.weight")
            if w.dtype not in (torch.uint8, torch.int8) or w.shape[1] != kb:
                raise ValueError(f"`{prefix}.{name}.weight`: expected packed int4 [N, {kb}] bytes, got {w.dtype} {tuple(w.shape)}")
            n = w.shape[0]
            s = take(f"{prefix}.{name}.weight_scales").reshape(-1)
            if s.numel() != n:
                raise ValueError(f"`{prefix}.{name}.weight_scales` has {s.numel()} entries for {n} rows")
            lin.weight[row:row + n].copy_(w.view(torch.int8) if w.dtype == torch.uint8 else w)
            lin.weight_scales.view(-1)[row:row + n].copy_(s.to(torch.float16))
            row += n
        if row != lin.weight.shape[0]:
            raise ValueError(f"{prefix}: {parts} give {row} rows, the fused projection has {lin.weight.shape[0]}")

    model.embed_tokens.copy_(take("model.embed_tokens.weight", model.embed_tokens.shape).to(torch.float16))
    model.lm_head.copy_(take("lm_head.weight", model.lm_head.shape).to(torch.float16))
    for i, layer in enumerate(model.layers):
        p = f"model.layers.{i}"
        put_linear(layer.qkv_proj, ("q_proj", "k_proj", "v_proj"), f"{p}.self_attn")      # quarot_llama.py:152-173
        put_linear(layer.o_proj, ("o_proj",), f"{p}.self_attn")
        put_linear(layer.gate_up, ("up_proj", "gate_proj"), f"{p}.mlp")                     # :301-314, up FIRST
        put_linear(layer.down_proj, ("down_proj",), f"{p}.mlp")
        hk = f"{p}.mlp.online_hadamard.had_rem_dim"
        if hk in sd:
            had = take(hk)
            if model.had_rem_dim is None or tuple(had.shape) != tuple(model.had_rem_dim.shape):
                raise ValueError(f"`{hk}` has shape {tuple(had.shape)}; the model's Hadamard factor is "
                                 f"{None if model.had_rem_dim is None else tuple(model.had_rem_dim.shape)}")
            # one shared factor in this engine: every layer must carry the same table (they do: get_hadK(I))
            h16 = had.to(torch.float16).to(model.had_rem_dim.device)
            if i == 0:
                model.had_rem_dim.copy_(h16)
            elif not torch.equal(h16, model.had_rem_dim):
                raise ValueError(f"`{hk}` differs from layer 0's table")
    if strict:
        extra = sorted(k for k in sd if k not in used and not k.endswith("rotary_emb.inv_freq"))
        if extra:
            raise ValueError(f"unexpected tensors in the checkpoint: {extra[:8]}{' ...' if len(extra) > 8 else ''}")
    return model


def load_qspec_checkpoint(model, model_dir: str, strict: bool = True):
    """model_runner.py:1132-1148 for the MI355X engine: read the shards, rename, fuse, in place."""
    return load_state_dict(model, read_state_dict(checkpoint_files(model_dir)), strict=strict)


@torch.no_grad()
def reference_state_dict(model) -> Dict[str, torch.Tensor]:
    """The model's weights under the reference's ON-DISK names (before the load-time renames), un-fused."""
    cfg = model.config
    q, kv, I = cfg.q_size, cfg.kv_size, cfg.intermediate_size
    out: Dict[str, torch.Tensor] = {
        "model.embed_tokens.weight": model.embed_tokens.detach().cpu().contiguous(),
        "lm_head.weight": model.lm_head.detach().cpu().contiguous(),
    }

    def split(lin, names_sizes, prefix, disk_name=lambda n: n):
        row = 0
        for name, n in names_sizes:
            out[f"{prefix}.{disk_name(name)}.weight"] = lin.weight[row:row + n].detach().cpu().view(torch.uint8).contiguous()
            out[f"{prefix}.{disk_name(name)}.weight_scales"] = lin.weight_scales.view(-1)[row:row + n].detach().cpu().view(-1, 1).contiguous()
            row += n

    for i, layer in enumerate(model.layers):
        p = f"model.layers.{i}"
        split(layer.qkv_proj, (("q_proj", q), ("k_proj", kv), ("v_proj", kv)), f"{p}.self_attn")
        split(layer.o_proj, (("o_proj.1", cfg.hidden_size),), f"{p}.self_attn")
        split(layer.gate_up, (("up_proj", I), ("gate_proj", I)), f"{p}.mlp")
        split(layer.down_proj, (("down_proj.2", cfg.hidden_size),), f"{p}.mlp")
        if model.had_rem_dim is not None:
            out[f"{p}.mlp.down_proj.0.had_rem_dim"] = model.had_rem_dim.detach().cpu().contiguous()
    return out


def save_qspec_checkpoint(model, model_dir: str, shards: int = 2):
    """Write `model-0000i-of-0000n.safetensors` in the reference's layout (two shards, as the reference expects)."""
    from safetensors.torch import save_file
    os.makedirs(model_dir, exist_ok=True)
    sd = reference_state_dict(model)
    keys = sorted(sd)
    per = (len(keys) + shards - 1) // shards
    paths = []
    for s in range(shards):
        part = {k: sd[k] for k in keys[s * per:(s + 1) * per]}
        path = os.path.join(model_dir, f"model-{s + 1:05d}-of-{shards:05d}.safetensors")
        save_file(part, path)
        paths.append(path)
    return paths
