"""Tensor-parallel verify path against the single-GPU path, all ranks on cuda:0.  Prints TP_OK on rank 0.

    torchrun --nproc-per-node 2 tests/tp_check.py --family tiny          (gloo ranks = processes; collectives host-staged)
    torchrun --nproc-per-node 2 tests/tp_check.py --family llama-2-13b   config 4: 40 heads (had40), I = 13824 (had108), TP = 2
    python tests/tp_check.py --family llama-3-70b --threads 8            config 5: TP = 8 with ranks as THREADS of one
                                                                          process (a GPU box admits <= 6 processes per card)

RCCL cannot put two ranks on one GPU; what is under test is the model wiring: K ranges of the row-parallel o_proj /
down_proj over the shared int4 buffer, channel ranges of gate_up, vocabulary ranges of lm_head, the exchange steps,
and that every rank ends a cycle on the same tokens.  Contract mirrored: RowParallelLinear all-reduce
(vllm/model_executor/layers/linear.py:1143), LogitsProcessor gather (logits_processor.py:104-107).
"""
import argparse
import os
import sys
import threading

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FAMILIES = {
    # hidden, intermediate, heads, kv heads, layers, vocab, rope theta, prompts, batch (k = 3)
    "tiny": (1024, 3584, 8, 2, 2, 2048, 10000.0, (20, 31, 8, 50)),
    "llama-2-13b": (5120, 13824, 40, 40, 1, 2048, 10000.0, (20, 31, 8, 50)),               # config 4: bs = 4
    "llama-3-70b": (8192, 28672, 64, 8, 1, 2048, 500000.0, (20, 31, 8, 50, 5, 17, 40, 9)),  # config 5: bs = 8
}


def run_rank(rank, world, comm, family, model_full, results):
    """One rank: (1) a verify forward TP vs single GPU on identical inputs, (2) engine cycles under TP."""
    import copy
    from qspec_amd.model import AttentionMetadata, QuarotLlamaConfig, Scratch
    from qspec_amd.parallel import TensorParallel
    from qspec_amd.spec_decode import QSpecEngine
    H, I, nh, nkv, L, V, theta, prompt_lens = FAMILIES[family]
    cfg = model_full.config
    dev = "cuda:0"
    torch.cuda.set_device(0)
    m = copy.copy(model_full)                 # same weight tensors, own TP context
    B, q_len, bs = len(prompt_lens), 4, 16
    rng = np.random.default_rng(3)
    ctx_lens = [int(c) for c in rng.integers(q_len + 1, 260, B)]
    ctx_lens[0] = q_len + 1
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
    for tp in (None, TensorParallel(rank, world, None, shard_layers=True, comm=comm)):
        m.tp = tp
        s = Scratch(cfg, T, B, q_len, 3, dev)
        kv = [(k.clone(), v.clone()) for k, v in kv0]
        hs = m.forward(ids, pos, kv, md, s, w4a4=False)
        logits = m.compute_logits(hs, s, shard_vocab=True)
        torch.cuda.synchronize()
        outs.append((hs.float().cpu().numpy().copy(), logits.float().cpu().numpy().copy(),
                     kv[-1][0].float().cpu().numpy().copy()))
    # Both paths against the fp64-accumulate CPU oracle (rank 0 computes it), then against each other.  Through a
    # layer's chain of fp16 roundings (see tests/test_model_gpu.py::test_teacher_forced_layers_llama3_8b) the normed
    # output is within 1e-3 for most elements and within 1e-2 everywhere; the TP path must be as close to the
    # oracle as the single-GPU path is (its partial sums are reduced in fp32 and rounded once).
    def rel(a, b, row_scale=False):
        den = np.maximum(1.0, np.abs(b))
        if row_scale:   # logits: a sum over the hidden size of products; an error is judged against the row's scale
            den = np.maximum(den, np.sqrt((b.astype(np.float64) ** 2).mean(axis=-1, keepdims=True)))
        return np.abs(a - b) / (1e-3 * den)
    stats = {}
    if rank == 0:
        import oracle as O
        from oracle.model import OracleModel
        O.build()
        om = OracleModel.from_torch_model(m, bs)
        kv_np = [(k.cpu().numpy().copy(), v.cpu().numpy().copy()) for k, v in kv0]
        ref = om.forward(ids.cpu().numpy(), pos.cpu().numpy(), kv_np, slots.cpu().numpy(), bt.cpu().numpy(),
                         np.array(ctx_lens, np.int32), (np.arange(B + 1) * q_len).astype(np.int32), False)
        ref_logits = om.logits(ref).astype(np.float32)
        for name, o in (("single", outs[0]), ("tp", outs[1])):
            rh, rl = rel(o[0], ref.astype(np.float32)), rel(o[1], ref_logits, True)
            stats[name] = (float(rh.max()), float((rh > 1).mean()), float(rl.max()), float((rl > 1).mean()))
        print("vs oracle (max err/1e-3, frac > 1e-3: hidden | logits): single %s, tp %s" % (stats["single"], stats["tp"]), flush=True)
        for name in ("single", "tp"):
            assert stats[name][0] < 10.0 and stats[name][1] < 0.1 and stats[name][2] < 10.0 and stats[name][3] < 0.25, (name, stats[name])
        assert stats["tp"][1] < 2.0 * stats["single"][1] + 0.01, stats      # TP no farther from the oracle than one GPU
    rh, rl = rel(outs[1][0], outs[0][0]), rel(outs[1][1], outs[0][1], True)
    print(f"rank {rank} tp vs single: hidden max {rh.max():.2f} frac {(rh > 1).mean():.3f}; logits max {rl.max():.2f} "
          f"frac {(rl > 1).mean():.3f}", flush=True)
    assert rh.max() < 15.0 and (rh > 1).mean() < 0.2, (rh.max(), (rh > 1).mean())
    assert rel(outs[1][2], outs[0][2]).max() < 10.0                     # last layer's KV (replicated attention side)
    # (2) the engine under TP: replicated draft + sharded verify must keep every rank on the same tokens
    m.tp = TensorParallel(rank, world, None, shard_layers=True, comm=comm)
    prng = np.random.default_rng(7)
    prompts = [prng.integers(0, cfg.vocab_size, n).tolist() for n in prompt_lens]
    eng = QSpecEngine(m, 3, B, max_model_len=128, block_size=16, max_new_tokens=32, use_graph=False, seed=5)
    eng.add_sequences(prompts)
    for _ in range(3):
        eng.step()
    gen = eng.generated()
    assert all(len(x) >= 4 for x in gen)
    assert eng.error_flag() == 0, "a device-side hand-off timed out"
    # (3) the draft pass's lm_head vocab-parallel as well (logit slices all-gathered): every logit is the same dot product
    # whichever rank streams its lm_head row, so the cycle must emit EXACTLY what it emits with the replicated draft head
    m.tp.shard_draft_vocab = True
    eng2 = QSpecEngine(m, 3, B, max_model_len=128, block_size=16, max_new_tokens=32, use_graph=False, seed=5)
    eng2.add_sequences(prompts)
    for _ in range(3):
        eng2.step()
    assert eng2.generated() == gen, "vocab-parallel draft lm_head changed the tokens"
    assert torch.equal(eng2.draft_probs_kbv, eng.draft_probs_kbv), "vocab-parallel draft lm_head changed the draft distributions"
    m.tp.shard_draft_vocab = False
    results[rank] = ([t for x in gen for t in x[:4]], float(rh.max()), float((rh > 1).mean()), float(rl.max()))


def build_model(family):
    from qspec_amd.model import QuarotLlamaConfig, QuarotLlamaForCausalLM
    H, I, nh, nkv, L, V, theta, _ = FAMILIES[family]
    cfg = QuarotLlamaConfig(H, I, nh, nkv, L, V, 1e-5, theta, 512, family)
    return QuarotLlamaForCausalLM(cfg, "cuda:0").init_synthetic(seed=1, lm_head_std=0.05)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--family", default="tiny", choices=list(FAMILIES))
    ap.add_argument("--threads", type=int, default=0, help="ranks as threads of this process (no torchrun)")
    a = ap.parse_args()
    if a.threads:
        from qspec_amd.parallel import ThreadComm
        world = a.threads
        model = build_model(a.family)
        shared = ThreadComm.Shared(world)
        results, errors = {}, []

        # Every rank thread gets a stream of its own: the per-(device, stream) workspaces of qspec_amd.ops (K-slice
        # partials of the long-K W4A16 GEMM, the exchange words of the spread kernels, ...) are then per rank, as they
        # are with one process per GPU.  (On ONE shared stream two ranks' launch pairs "write workspace | consume
        # workspace" interleave: a rank then read another rank's partials, about one run in ten.)  As in the
        # process-mode rehearsal below, the ranks' kernels compete for one GPU's CUs, so the one-workgroup-per-token
        # forms are used.
        os.environ.setdefault("QSPEC_XWG_SPREAD", "0")

        def body(r):
            try:
                with torch.cuda.stream(torch.cuda.Stream(device="cuda:0")):
                    run_rank(r, world, ThreadComm(shared, r), a.family, model, results)
            except BaseException as exc:  # noqa: BLE001 -- a dead rank would leave the others in the barrier
                errors.append((r, repr(exc)))
                shared.barrier.abort()
        ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errors, errors
        assert all(results[r][0] == results[0][0] for r in range(world)), "ranks diverged"
        print("TP_OK %s world=%d (threads) max_hidden_err/1e-3=%.2f frac>1e-3=%.2e max_logit_err/1e-3=%.2f tokens=%s"
              % (a.family, world, results[0][1], results[0][2], results[0][3], results[0][0][:4]), flush=True)
        return
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # Ranks as PROCESSES share one GPU here, so their kernels compete for its CUs: that voids the residency the
    # kernels that spread a token over several workgroups rely on (at most one workgroup per CU of an otherwise idle
    # GPU: all partners of a token resident while they wait for each other).  A TP deployment gives every rank its own
    # GPU; for this host-staged rehearsal the one-workgroup-per-token forms are used (same bytes, tested elsewhere).
    os.environ.setdefault("QSPEC_XWG_SPREAD", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    results = {}
    run_rank(rank, world, None, a.family, build_model(a.family), results)
    mine = torch.tensor(results[rank][0], dtype=torch.int64)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    assert all(torch.equal(gathered[0], g) for g in gathered), "ranks diverged"
    if rank == 0:
        print("collectives:", "oneshot+gloo" if os.environ.get("QSPEC_ONESHOT_AR") == "1" else "gloo (host staged)", flush=True)
        print("TP_OK %s world=%d max_hidden_err/1e-3=%.2f frac>1e-3=%.2e max_logit_err/1e-3=%.2f tokens=%s"
              % (a.family, world, results[0][1], results[0][2], results[0][3], results[0][0][:4]), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
