"""Checkpoint loader for the reference's on-disk format (model_runner.py:1132-1148): runs on CPU tensors, no kernels."""
import os

import pytest
import torch

from qspec_amd import checkpoint
from qspec_amd.model import QuarotLlamaConfig, QuarotLlamaForCausalLM


def tiny_cfg():
    return QuarotLlamaConfig(1024, 3584, 8, 2, 2, 512, 1e-5, 10000.0, 256, "tiny")


def make(seed):
    m = QuarotLlamaForCausalLM(tiny_cfg(), "cpu")
    g = torch.Generator().manual_seed(seed)
    for layer in m.layers:
        for lin in layer.linears():
            n, kb = lin.weight.shape
            lin.weight.copy_(torch.randint(0, 256, (n, kb), generator=g, dtype=torch.int16).to(torch.uint8).view(torch.int8))
            lin.weight_scales.copy_((torch.rand(n, 1, generator=g) * 0.01 + 1e-3).to(torch.float16))
    m.embed_tokens.copy_((torch.randn(m.embed_tokens.shape, generator=g) * 0.02).to(torch.float16))
    m.lm_head.copy_((torch.randn(m.lm_head.shape, generator=g) * 0.02).to(torch.float16))
    return m


def test_reference_names_and_rename_rule():
    sd = checkpoint.reference_state_dict(make(0))
    # on-disk names of the reference checkpoint: Sequential indices on o_proj / down_proj, separate q/k/v and up/gate
    for k in ("model.layers.0.self_attn.q_proj.weight", "model.layers.0.self_attn.o_proj.1.weight_scales",
              "model.layers.1.mlp.down_proj.2.weight", "model.layers.1.mlp.down_proj.0.had_rem_dim",
              "model.layers.0.mlp.up_proj.weight", "model.layers.0.mlp.gate_proj.weight_scales",
              "model.embed_tokens.weight", "lm_head.weight"):
        assert k in sd, k
    assert sd["model.layers.0.self_attn.q_proj.weight"].dtype == torch.uint8
    assert tuple(sd["model.layers.0.self_attn.k_proj.weight_scales"].shape) == (256, 1)
    assert checkpoint._rename("model.layers.3.mlp.down_proj.0.had_rem_dim") == "model.layers.3.mlp.online_hadamard.had_rem_dim"
    assert checkpoint._rename("model.layers.3.mlp.down_proj.2.weight") == "model.layers.3.mlp.down_proj.weight"
    assert checkpoint._rename("model.layers.3.self_attn.o_proj.1.weight") == "model.layers.3.self_attn.o_proj.weight"


def test_save_load_round_trip_and_fusion_order(tmp_path):
    src = make(1)
    paths = checkpoint.save_qspec_checkpoint(src, str(tmp_path))
    assert [os.path.basename(p) for p in paths] == ["model-00001-of-00002.safetensors", "model-00002-of-00002.safetensors"]
    dst = QuarotLlamaForCausalLM(tiny_cfg(), "cpu")
    checkpoint.load_qspec_checkpoint(dst, str(tmp_path))
    for a, b in zip(src.layers, dst.layers):
        for la, lb in zip(a.linears(), b.linears()):
            assert torch.equal(la.weight, lb.weight) and torch.equal(la.weight_scales, lb.weight_scales)
    assert torch.equal(src.embed_tokens, dst.embed_tokens) and torch.equal(src.lm_head, dst.lm_head)
    assert torch.equal(src.had_rem_dim, dst.had_rem_dim)
    # fusion order of the reference: qkv = [q; k; v], gate_up = [up; gate]  (quarot_llama.py:152-173,301-314)
    sd = checkpoint.read_state_dict(paths)
    cfg = src.config
    l0 = dst.layers[0]
    assert torch.equal(l0.qkv_proj.weight[cfg.q_size:cfg.q_size + cfg.kv_size].view(torch.uint8),
                       sd["model.layers.0.self_attn.k_proj.weight"])
    assert torch.equal(l0.gate_up.weight[:cfg.intermediate_size].view(torch.uint8), sd["model.layers.0.mlp.up_proj.weight"])
    assert torch.equal(l0.gate_up.weight_scales.view(-1)[cfg.intermediate_size:],
                       sd["model.layers.0.mlp.gate_proj.weight_scales"].view(-1))


def test_loader_rejects_bad_checkpoints(tmp_path):
    src = make(2)
    sd = {checkpoint._rename(k): v for k, v in checkpoint.reference_state_dict(src).items()}
    dst = QuarotLlamaForCausalLM(tiny_cfg(), "cpu")
    missing = dict(sd)
    del missing["model.layers.1.mlp.down_proj.weight"]
    with pytest.raises(KeyError):
        checkpoint.load_state_dict(dst, missing)
    wrong = dict(sd)
    wrong["model.layers.0.self_attn.q_proj.weight"] = wrong["model.layers.0.self_attn.q_proj.weight"][:, :-1].contiguous()
    with pytest.raises(ValueError):
        checkpoint.load_state_dict(dst, wrong)
    extra = dict(sd)
    extra["model.layers.0.self_attn.surprise"] = torch.zeros(1)
    with pytest.raises(ValueError):
        checkpoint.load_state_dict(dst, extra)
    checkpoint.load_state_dict(dst, extra, strict=False)
    with pytest.raises(FileNotFoundError):
        checkpoint.load_qspec_checkpoint(dst, str(tmp_path / "nothing_here"))
