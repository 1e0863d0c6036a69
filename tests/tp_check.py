"""Run under torchrun (2 ranks, gloo, both on cuda:0): the tensor-parallel verify path of the engine against the
single-GPU engine on the tiny model.  Prints TP_OK on rank 0.  (RCCL cannot put two ranks on one GPU; the
collectives are staged through the host here, the model wiring is what is under test.)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qspec_amd.model import QuarotLlamaConfig, QuarotLlamaForCausalLM
    from qspec_amd.parallel import TensorParallel
    from qspec_amd.spec_decode import QSpecEngine
    cfg = QuarotLlamaConfig(1024, 3584, 8, 2, 2, 2048, 1e-5, 10000.0, 512, "tiny")
    dev = "cuda:0"
    rng = np.random.default_rng(7)
    prompts = [rng.integers(0, cfg.vocab_size, n).tolist() for n in (20, 31, 8, 50)]
    # (1) one verify forward on identical inputs: TP (K / channel / vocab shards + collectives) vs single GPU
    from qspec_amd.model import AttentionMetadata, Scratch
    m = QuarotLlamaForCausalLM(cfg, dev).init_synthetic(seed=1, lm_head_std=0.05)
    B, q_len, bs = 4, 4, 16
    ctx_lens = [40, 130, 9, 260]
    max_blocks = max((c + bs - 1) // bs for c in ctx_lens) + 1
    nb = B * max_blocks
    g = torch.Generator().manual_seed(3)
    bt = torch.arange(nb, dtype=torch.int32).view(B, max_blocks).to(dev)
    shape = (nb, bs, cfg.num_key_value_heads, cfg.head_dim)
    kv0 = [((torch.randn(shape, generator=g) * 0.5).half().to(dev), (torch.randn(shape, generator=g) * 0.5).half().to(dev))
           for _ in range(cfg.num_hidden_layers)]
    T = B * q_len
    ids = torch.randint(0, cfg.vocab_size, (T,), generator=g).to(dev)
    pos = torch.cat([torch.arange(c - q_len, c) for c in ctx_lens]).to(dev)
    slots = torch.cat([bt[b].cpu().long()[torch.arange(c - q_len, c) // bs] * bs + torch.arange(c - q_len, c) % bs
                       for b, c in enumerate(ctx_lens)]).to(dev)
    md = AttentionMetadata(slots, bt, torch.tensor(ctx_lens, dtype=torch.int32, device=dev),
                           (torch.arange(B + 1, dtype=torch.int32) * q_len).to(dev), q_len, 3)
    outs = []
    for tp in (None, TensorParallel(rank, world, None)):
        m.tp = tp
        s = Scratch(cfg, T, B, q_len, 3, dev)
        kv = [(k.clone(), v.clone()) for k, v in kv0]
        hs = m.forward(ids, pos, kv, md, s, w4a4=False)
        logits = m.compute_logits(hs, s, shard_vocab=True)
        torch.cuda.synchronize()
        outs.append((hs.float().cpu().numpy().copy(), logits.float().cpu().numpy().copy(), kv[1][0].cpu().numpy().copy()))
    dh = np.abs(outs[0][0] - outs[1][0])
    dl = np.abs(outs[0][1] - outs[1][1])
    assert dh.max() < 3e-2 and np.median(dh) < 2e-3, (dh.max(), np.median(dh))
    assert dl.max() < 5e-2, dl.max()
    assert np.abs(outs[0][2].astype(np.float32) - outs[1][2].astype(np.float32)).max() < 3e-2   # layer-1 KV
    tv = dh
    # (2) the engine under TP: replicated draft + all-reduced verify must keep every rank on the same tokens
    res = []
    for tp_world in (world,):
        m = QuarotLlamaForCausalLM(cfg, dev).init_synthetic(seed=1, lm_head_std=0.05)
        m.tp = TensorParallel(rank, world, None)
        eng = QSpecEngine(m, 3, 4, max_model_len=256, block_size=16, max_new_tokens=64, use_graph=False, seed=5)
        eng.add_sequences(prompts)
        for _ in range(4):
            eng.step()
        res.append((eng.generated(), eng.metrics()))
    mine = torch.tensor([t for g in res[0][0] for t in g[:6]], dtype=torch.int64)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    assert all(torch.equal(gathered[0], g) for g in gathered), "ranks diverged"
    assert all(len(g) >= 5 for g in res[0][0])
    if rank == 0:
        print("TP_OK max_hidden_diff=%.3e tokens=%s" % (tv.max(), res[0][0][0][:6]), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
