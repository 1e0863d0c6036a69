#!/usr/bin/env python3
"""Decode-only workload for rocprofv3: the QSpec cycle graph replayed N times with NO prefill in the process
(the KV cache is filled with random values and the sequence state is set directly), so every kernel in the
trace belongs to the timed region of bench.py.

    rocprofv3 --kernel-trace --stats -d gpurun_out/profX -- python3 scripts/profile_cycle.py --steps 20
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--model", default="llama-3-8b")
    p.add_argument("--k", type=int, default=3)
    p.add_argument("--batch", type=int, default=4)
    p.add_argument("--ctx", type=int, default=512)
    p.add_argument("--agreement", type=float, default=0.96)
    p.add_argument("--eager", action="store_true")
    p.add_argument("--sync-every-step", action="store_true",
                   help="drain the GPU after every cycle.  Needed under `rocprofv3 --pmc`: the counter-collection tool faults "
                        "(SIGSEGV in one of its threads) once roughly 8 k profiled dispatches are outstanding, i.e. with a few "
                        "cycle graphs (1.0 k launches at bs=4, 2.3 k at bs=32) enqueued back to back without a host sync")
    p.add_argument("--plain-engine", action="store_true", help="the product engine, no synthetic-agreement hook (no bench library loaded)")
    a = p.parse_args()

    def stage(msg):   # progress marks on stderr: where a profiled run was when a tool-side fault took the process down
        print(f"[profile_cycle] {msg}", file=sys.stderr, flush=True)
    import faulthandler
    faulthandler.enable()
    from qspec_amd.model import CONFIGS, QuarotLlamaForCausalLM
    if a.plain_engine:
        from qspec_amd.spec_decode import QSpecEngine
    else:
        import bench
        QSpecEngine = bench.make_bench_engine_class()
    dev = "cuda:0"
    cfg = CONFIGS[a.model]
    model = QuarotLlamaForCausalLM(cfg, dev).init_synthetic(0, 0.02)
    torch.cuda.synchronize()
    stage("model built")
    total = a.steps + 8
    eng = QSpecEngine(model, a.k, a.batch, max_model_len=a.ctx + total * (a.k + 1) + 32, block_size=16,
                      max_new_tokens=total * (a.k + 1) + 8, use_graph=not a.eager, seed=0)
    if not a.plain_engine:
        eng.set_agreement(a.agreement)
    g = torch.Generator(device=dev).manual_seed(1)
    for kc, vc in eng.kv_caches:
        kc.copy_((torch.randn(kc.shape, generator=g, device=dev) * 0.5).half())
        vc.copy_((torch.randn(vc.shape, generator=g, device=dev) * 0.5).half())
    eng.seq_lens.fill_(a.ctx + 1)
    eng.last_token.copy_(torch.randint(0, cfg.vocab_size, (a.batch,), generator=g, device=dev))
    eng.gen_lens.fill_(1)
    eng._len_ub = [a.ctx + 1] * a.batch
    eng._gen_ub = [1] * a.batch
    eng.n_active = a.batch
    stage("engine built, state set")
    for i in range(3):
        eng.step()
        torch.cuda.synchronize()
        stage(f"warm-up step {i} done (graph={'yes' if eng._graph is not None else 'no'})")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.step()
        if a.sync_every_step:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"cycle_ms={dt / a.steps * 1e3:.3f} metrics={eng.metrics()}", flush=True)


if __name__ == "__main__":
    main()
