#!/usr/bin/env python3
"""In-graph time of the non-GEMM launches of a draft layer (attention, merge + head Hadamard, MLP Hadamard) on the engine's
own buffers and metadata: a hipGraph of one launch per layer (32 different KV caches), replayed; HIP events.  Dev tool.

    python3 scripts/bench_layer_parts.py [--batch 4] [--ctx 512]
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def timeit(body, reps=10):
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--batch", type=int, default=4)
    p.add_argument("--ctx", type=int, default=512)
    a = p.parse_args()
    from qspec_amd import ops
    from qspec_amd.model import CONFIGS, QuarotLlamaForCausalLM
    from qspec_amd.spec_decode import QSpecEngine
    dev = "cuda:0"
    cfg = CONFIGS["llama-3-8b"]
    model = QuarotLlamaForCausalLM(cfg, dev).init_synthetic(0, 0.02)
    eng = QSpecEngine(model, 3, a.batch, max_model_len=a.ctx + 256, block_size=16, max_new_tokens=128, use_graph=True, seed=0)
    g = torch.Generator(device=dev).manual_seed(1)
    for kc, vc in eng.kv_caches:
        kc.copy_((torch.randn(kc.shape, generator=g, device=dev) * 0.5).half())
        vc.copy_((torch.randn(vc.shape, generator=g, device=dev) * 0.5).half())
    eng.seq_lens.fill_(a.ctx + 1); eng.gen_lens.fill_(1)
    eng.last_token.copy_(torch.randint(0, cfg.vocab_size, (a.batch,), generator=g, device=dev))
    eng._len_ub = [a.ctx + 1] * a.batch; eng._gen_ub = [1] * a.batch; eng.n_active = a.batch
    for _ in range(2):
        eng.step()
    torch.cuda.synchronize()
    s, md, B = eng.scratch_draft, eng.md_draft, a.batch
    nh, hd = cfg.num_attention_heads, cfg.head_dim
    row = cfg.q_size + 2 * cfg.kv_size
    qkv = s.act_buffer_qkv[:B]
    L = len(model.layers)

    def attn():
        for kc, vc in eng.kv_caches:
            ops.paged_attention(qkv, row, kc, vc, md.block_tables, md.ctx_lens, md.q_start, B, md.max_q_len, nh, model.sm_scale,
                                md.n_splits, s.attn_ws, None)

    def merge_spread():
        for _ in range(L):
            ops.heads_hadamard_merged_spread(s.attn_ws, B * md.max_q_len, md.n_splits, B, nh, hd, model.head_had_scale,
                                             s.act_buffer_had[:B], s.had_part_amax[:B])

    def merge_one():
        for _ in range(L):
            ops.heads_hadamard_merged(s.attn_ws, B * md.max_q_len, md.n_splits, B, nh, hd, model.head_had_scale,
                                      q=s.quantized_buffer_qkv[:B], scale=s.scale_buffer[:B])

    def mlp_had():
        for _ in range(L):
            ops.mlp_hadamard(s.act_buffer_had_mlp[:B], model.had_rem_dim, model.had_K, model.mlp_had_scale,
                             q=s.quantized_buffer_mlp[:B], scale=s.scale_buffer[:B])

    for name, f in (("attention (8 splits, partials)", attn), ("merge + head Hadamard, spread (fp16 + maxima)", merge_spread),
                    ("merge + head Hadamard + quant, one workgroup per token", merge_one), ("MLP Hadamard + quant", mlp_had)):
        print(f"{name}: {timeit(f) / L:.2f} us per launch (graph of {L} launches, boundary included)", flush=True)


if __name__ == "__main__":
    main()
