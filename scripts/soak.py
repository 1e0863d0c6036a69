#!/usr/bin/env python3
"""Soak run of the product engine: the captured draft -> verify -> accept cycle replayed for minutes while requests of random
prompt lengths finish and join (the prompt pass between replays, never a re-capture), a share of them sampled instead of greedy.
Per cycle the worker's one host read (out tokens + error word) is checked; at the end: no recovery was needed, no sticky error
word is set, every emitted token is a vocabulary id, the sequence lengths on the device equal the host's bookkeeping, and the
cycle time of the last tenth of the run is compared with the first tenth's (drift).  Prints one JSON line.

    python scripts/soak.py --minutes 10            # Llama-3-8B shapes, k = 3, 4 slots
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--minutes", type=float, default=5.0)
    p.add_argument("--model", default="llama-3-8b")
    p.add_argument("--k", type=int, default=3)
    p.add_argument("--batch", type=int, default=4)
    p.add_argument("--max-new", type=int, default=192)
    p.add_argument("--sampled-share", type=float, default=0.25)
    p.add_argument("--seed", type=int, default=0)
    a = p.parse_args()
    from qspec_amd.model import CONFIGS, QuarotLlamaForCausalLM
    from qspec_amd.spec_decode import QSpecEngine
    dev = "cuda:0"
    cfg = CONFIGS[a.model]
    V = cfg.vocab_size
    model = QuarotLlamaForCausalLM(cfg, dev).init_synthetic(a.seed, 0.02)
    max_prompt = 512
    eng = QSpecEngine(model, a.k, a.batch, max_model_len=max_prompt + a.max_new + 4 * (a.k + 1) + 16, block_size=16,
                      max_new_tokens=a.max_new + 4 * (a.k + 1), use_graph=True, seed=a.seed)
    rng = np.random.default_rng(a.seed)
    budget = [0] * a.batch          # tokens a slot's request may still emit
    emitted = [0] * a.batch

    def admit(slot):
        n = int(rng.integers(8, max_prompt + 1))
        eng.add_sequences_to([slot], [rng.integers(0, V, n).tolist()])
        if rng.random() < a.sampled_share:
            eng.set_sampling_params(slot, temperature=float(rng.uniform(0.5, 1.2)), top_k=int(rng.choice([-1, 20, 200])),
                                    top_p=float(rng.choice([1.0, 0.9])))
        budget[slot] = int(rng.integers(16, a.max_new))
        emitted[slot] = 0

    for b in range(a.batch):
        admit(b)
    t_end = time.perf_counter() + a.minutes * 60.0
    cycles = requests = tokens = bad_tokens = errors = 0
    times = []
    graphs = set()
    last_report = time.perf_counter()
    while time.perf_counter() < t_end:
        t0 = time.perf_counter()
        eng.step()
        out, err = eng.read_outputs()
        times.append(time.perf_counter() - t0)
        cycles += 1
        graphs.add(id(eng._graph_s if eng._mode_sampling else eng._graph))
        if err:
            errors += 1
            eng.recover()
            out, err = eng.read_outputs()
            if err:
                raise SystemExit(f"cycle {cycles}: error word {err} after recovery")
        o = out.numpy()
        eng.note_emitted([int((o[b] >= 0).sum()) for b in range(a.batch)])   # exact lengths, as the worker reports them
        for b in range(a.batch):
            row = o[b][o[b] >= 0]
            bad_tokens += int((row >= V).sum())
            emitted[b] += len(row)
            tokens += len(row)
            if emitted[b] >= budget[b]:           # the request is done: its slot goes to a new one
                eng.free_slot(b)
                admit(b)
                requests += 1
        if time.perf_counter() - last_report > 60.0:
            print(f"[soak] {cycles} cycles, {requests} requests, {tokens} tokens", file=sys.stderr, flush=True)
            last_report = time.perf_counter()
    torch.cuda.synchronize()
    eng.sync_lens()
    lens_ok = eng._len_ub == eng.seq_lens.tolist()
    tenth = max(1, len(times) // 10)
    first, last = float(np.median(times[tenth // 2:tenth])), float(np.median(times[-tenth:]))
    res = {"model": a.model, "k": a.k, "batch": a.batch, "minutes": a.minutes, "cycles": cycles, "requests_completed": requests,
           "tokens_emitted": tokens, "tokens_outside_vocabulary": bad_tokens, "error_words": errors, "recoveries": eng.recoveries,
           "sticky_error_flag": eng.error_flag(), "device_lengths_match_host": bool(lens_ok),
           "graphs_seen": len(graphs), "cycle_ms_incl_host_read_median_first_tenth": round(first * 1e3, 3),
           "cycle_ms_incl_host_read_median_last_tenth": round(last * 1e3, 3),
           "cycle_ms_p99": round(float(np.quantile(times, 0.99)) * 1e3, 3),
           "hbm_allocated_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2)}
    print(json.dumps(res), flush=True)
    ok = bad_tokens == 0 and errors == 0 and eng.recoveries == 0 and res["sticky_error_flag"] == 0 and lens_ok
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
