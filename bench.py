#!/usr/bin/env python3
"""bench.py -- accepted tokens/s + draft-accept-rate of the QSpec draft/verify cycle on MI355X.

    python bench.py --gpus 1 --steps 100 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one speculative cycle of the whole batch: k W4A4 draft forwards + one W4A16 verify forward over
k+1 tokens per sequence + rejection sampling + commit (one hipGraph replay).  Workload = BASELINE.json
configs[1]: Llama-3-8B QSpec, k=3, bs=4, synthetic weights and prompts.  Prefill is outside
the timed region (inputs resident in HBM when timing starts).

One JSON line on rank 0.  The hardware quantity is `ms_per_step` (one cycle; identical with the synthetic knob on or
off): the --steps region is timed `--repeats` (3) times, every repeat from the SAME sequence state (context lengths, tokens, RNG
state restored: the repeats are the same work), and the MEDIAN region is reported, every region listed in `ms_per_step_repeats`.  With N = 1 and the default workload the line also carries `other_configs`: the other BASELINE.json
configs' shapes (k=5 bs=32; Llama-2-13B; Llama-3-70B; TinyLlama) built, timed the same way and freed one after the other on
this GPU, each with its own `ms_per_step`, `value` and dominant-kernel `roofline`.  `value` = emitted tokens per second of the whole job at the SYNTHETIC draft/target agreement named in
`config.agreement` (random int4 weights agree ~1-4 %; the reference's trained checkpoint 0.96; SURVEY.md 8d
sanctions the controlled-agreement mode) -- the same cycles at the weights' own agreement are under `natural_agreement`,
and `e2e_incl_prefill` is the demo.py:139-160 figure (prompt pass + decode to max_tokens, tokens / wall time).  Plus
  roofline      dominant kernel (the W4A4 weight-streaming GEMM): algorithmic bytes / measured launch time
  cpu_baseline  the CPU oracle ("port" of the reference arithmetic) timed on this host's cores on a bounded sample

The agreement knob is bench-only code: BenchEngine below + bench_kernels/libqspec_bench.so; nothing of it is in the
product library or its header.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--model", default="llama-3-8b")
    p.add_argument("--k", type=int, default=3)
    p.add_argument("--batch", type=int, default=4)
    p.add_argument("--prompt-len", type=int, default=512)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--lm-head-std", type=float, default=0.02)
    p.add_argument("--agreement", default="0.96",
                   help="synthetic draft/target agreement rate of the headline run (random weights agree ~4%%, the "
                        "reference's trained checkpoint 0.96, BASELINE.md); 'none' = the weights' own agreement")
    p.add_argument("--natural-steps", type=int, default=30, help="extra cycles measured with the weights' own agreement")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--cpu-layers", type=int, default=4, help="layers of the model the CPU baseline sample runs")
    p.add_argument("--cpu-cycles", type=int, default=5, help="timed CPU cycles (median), after --cpu-warmup untimed ones")
    p.add_argument("--cpu-warmup", type=int, default=2)
    p.add_argument("--e2e-max-tokens", type=int, default=256, help="max_tokens of the end-to-end (prefill included) run; 0 = skip")
    p.add_argument("--repeats", type=int, default=3, help="the --steps region is timed this many times back to back; "
                   "ms_per_step / value are the MEDIAN repeat, every repeat is listed beside them")
    p.add_argument("--other-configs", default="auto",
                   help="'auto' (N = 1, default workload only): after the headline also time the other BASELINE.json configs' shapes "
                        "on this GPU and report them under other_configs; 'none' = skip; or a list 'model:k:batch,...'")
    p.add_argument("--other-steps", type=int, default=20)
    return p.parse_args()


# BASELINE.json configs other than the headline, as (model, k, batch, label).  Configs 4 / 5 are quoted at TP = 2 / 8; the draft
# pass (3 of the 4 forwards of a cycle) runs replicated under this design, so their shapes on ONE GPU are what every rank runs
# for the draft pass and an upper bound for the verify pass.
OTHER_CONFIGS = (("llama-3-8b", 5, 32, "config 3: Llama-3-8B k=5 bs=32 TP=1"),
                 ("llama-2-13b", 3, 4, "config 4 shapes on one GPU: Llama-2-13B k=3 bs=4 (quoted at TP=2)"),
                 ("llama-3-70b", 3, 8, "config 5 shapes on one GPU: Llama-3-70B k=3 bs=8 (quoted at TP=8)"),
                 ("tinyllama-1.1b", 3, 1, "config 1 shapes on the HIP path: TinyLlama-1.1B k=3 bs=1 (quoted on the CPU executor)"))


def make_bench_engine_class():
    """QSpecEngine + the synthetic-agreement knob (bench only; see module docstring)."""
    import ctypes
    from qspec_amd.spec_decode import QSpecEngine
    path = os.path.join(ROOT, "bench_kernels", "libqspec_bench.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "bench_kernels"), "-s", "libqspec_bench.so"])
    lib = ctypes.CDLL(path)
    fn = lib.qspec_bench_force_agreement
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p,
                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

    class BenchEngine(QSpecEngine):
        _rho = None

        def set_agreement(self, rho):
            self._rho = rho

        # the sequence state at the start of the timed regions: every repeat of the --steps region starts from the SAME
        # state (same context lengths, same tokens, same RNG state), so the repeats time the same work -- without it the
        # context grows by ~(k+1) tokens per cycle and a later region is a different (longer-context) workload
        def save_region_state(self):
            return ([t.clone() for t in (self.seq_lens, self.gen_lens, self.last_token, self.sampler.rng_state)],
                    list(self._len_ub), list(self._gen_ub))

        def restore_region_state(self, state):
            tensors, len_ub, gen_ub = state
            for t, s in zip((self.seq_lens, self.gen_lens, self.last_token, self.sampler.rng_state), tensors):
                t.copy_(s)
            self._len_ub, self._gen_ub = list(len_ub), list(gen_ub)

        def _verify_logits_hook(self, draft_ids):
            if self._rho is None:
                return None
            rho, rng = float(self._rho), self.sampler.rng_state

            def hook(logits):
                B, k = draft_ids.shape
                rc = fn(logits.data_ptr(), draft_ids.data_ptr(), draft_ids.stride(0), draft_ids.stride(1), rho,
                        rng.data_ptr(), B, k, logits.shape[-1], torch.cuda.current_stream().cuda_stream)
                assert rc == 0
            return hook
    return BenchEngine


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run ... bench.py <same flags>` as
    a CHILD process -- before this process has made any GPU call (never exec from a process that has initialised the
    GPU) -- relay its output (rank 0 prints the JSON line) and exit with its code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def dist_setup(n):
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        import torch.distributed as dist
        # QSPEC_BENCH_BACKEND=gloo: rehearsal of the N > 1 harness on a box with fewer GPUs than ranks (ranks share
        # devices round-robin, collectives staged through the host); the measured configuration is always RCCL
        backend = os.environ.get("QSPEC_BENCH_BACKEND", "nccl")
        if backend != "nccl":
            local = local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)
    assert world == n, f"--gpus {n} but WORLD_SIZE={world}: launch with torch.distributed.run"
    return rank, world, local


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def measure_dominant_kernel(model, engine, reps=5):
    """The dominant kernel = the weight-streaming W4A4 GEMM of the draft pass (qspec::gemm_w4a4_stream_kernel: the
    four decoder GEMM launches of a draft forward at M = batch, 3 of the 4 forwards of a cycle), in exactly the
    forms the cycle launches: LN + int4 quant prologue -> qkv (+ RoPE + KV write) / gate_up (+ silu*up), and the
    (xq, xs) form with the residual add in the epilogue for o_proj / down_proj.  Every decoder GEMM of one draft forward is enqueued in model
    order on the real weights into one hipGraph (launches back to back as in the timed region; each weight byte is
    cold again by the time it is re-read: 3.5 GB > the 256 MiB infinity cache), bracketed by HIP events recorded on
    the launching stream.  Returns algorithmic bytes and seconds per launch, overall and per shape."""
    from qspec_amd import ops
    B = engine.B
    s, md, cfg = engine.scratch_draft, engine.md_draft, model.config
    xq, sc = s.quantized_buffer_qkv[:B], s.scale_buffer[:B]
    hid, o = s.hidden[:B], s.act_buffer_output[:B]
    eps = cfg.rms_norm_eps
    kinds = ("qkv", "o", "gate_up", "down")
    # the same dispatch as QuarotLlamaForCausalLM.forward: fused epilogues need head_dim 128, fused norms M <= 16
    fuse = cfg.head_dim == 128
    ln_fused = (fuse and ops.ln_linear_s4s4_supported(B, cfg.q_size + 2 * cfg.kv_size, cfg.hidden_size)
                and ops.ln_linear_s4s4_supported(B, 2 * cfg.intermediate_size, cfg.hidden_size)
                and ops.rowwise_scaled_linear_s4s4_residual_supported(B, cfg.hidden_size, cfg.hidden_size)
                and ops.rowwise_scaled_linear_s4s4_residual_supported(B, cfg.hidden_size, cfg.intermediate_size))
    gu_out = s.act_buffer_gate_up[:B]
    # o_proj at <= 4 tokens: fp16 head-Hadamard rows + partial row maxima, quantised in the launch's prologue (model.py)
    hq = (ln_fused and model.HADAMARD_QUANT_IN_OPROJ and model.MERGE_IN_HADAMARD
          and (ops.heads_hadamard_merged_spread_supported(B, cfg.num_attention_heads, cfg.head_dim) if model.head_had_K == 1 else
               ops.heads_hadamard_mix_merged_spread_supported(B, cfg.num_attention_heads, cfg.head_dim, model.head_had_K))
          and ops.rowwise_scaled_linear_s4s4_residual_hq_supported(B, cfg.hidden_size, cfg.q_size))

    def launch(layer, kc, vc, kind):
        if kind == "qkv":
            if ln_fused:
                ops.ln_qkv_rope_linear(hid, None, None, eps, layer.qkv_proj.weight, layer.qkv_proj._scales(),
                                       s.act_buffer_qkv[:B], engine.d_pos, model.cos_sin_cache, kc, vc, md.slot_mapping,
                                       cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim)
            elif fuse:
                ops.qkv_rope_linear(xq, sc, layer.qkv_proj.weight, layer.qkv_proj._scales(), s.act_buffer_qkv[:B],
                                    engine.d_pos, model.cos_sin_cache, kc, vc, md.slot_mapping, cfg.num_attention_heads,
                                    cfg.num_key_value_heads, cfg.head_dim)
            else:
                ops.rowwise_scaled_linear_cutlass_s4s4_unified(xq, sc, layer.qkv_proj.weight, layer.qkv_proj._scales(), None,
                                                               s.act_buffer_qkv[:B])
        elif kind == "o":
            if hq:
                ops.rowwise_scaled_linear_s4s4_residual_hq(s.act_buffer_had[:B], s.had_part_amax[:B], 1.0, layer.o_proj.weight,
                                                           layer.o_proj._scales(), hid, hid)
            elif ln_fused:
                ops.rowwise_scaled_linear_s4s4_residual(xq, sc, layer.o_proj.weight, layer.o_proj._scales(), hid, hid)
            else:
                ops.rowwise_scaled_linear_cutlass_s4s4_unified(xq, sc, layer.o_proj.weight, layer.o_proj._scales(), None, o)
        elif kind == "gate_up":
            if ln_fused:
                ops.ln_gate_up_silu_linear(hid, None, None, eps, layer.gate_up.weight, layer.gate_up._scales(),
                                           s.act_buffer_had_mlp[:B])
            elif fuse:
                ops.gate_up_silu_linear(xq, sc, layer.gate_up.weight, layer.gate_up._scales(), s.act_buffer_had_mlp[:B])
            else:
                ops.rowwise_scaled_linear_cutlass_s4s4_unified(xq, sc, layer.gate_up.weight, layer.gate_up._scales(), None, gu_out)
        elif ln_fused:
            ops.rowwise_scaled_linear_s4s4_residual(s.quantized_buffer_mlp[:B], sc, layer.down_proj.weight, layer.down_proj._scales(), hid, hid)
        else:
            ops.rowwise_scaled_linear_cutlass_s4s4_unified(s.quantized_buffer_mlp[:B], sc, layer.down_proj.weight, layer.down_proj._scales(), None, o)

    def nbytes(kd, lin):
        n, kb = lin.weight.shape
        out_cols = n // 2 if kd == "gate_up" else n
        w = n * kb + 2 * n                                     # packed weights + channel scales
        if ln_fused and kd in ("qkv", "gate_up"):
            act_in = B * (2 * kb) * 2                          # the residual stream, fp16 (normed in the prologue)
        elif hq and kd == "o":
            act_in = B * (2 * kb) * 2 + B * 8 * 4              # fp16 rows + partial maxima (quantised in the prologue)
        else:
            act_in = B * kb + 2 * B                            # packed int4 activations + scales
        if ln_fused and kd in ("o", "down"):
            act_in += 2 * B * out_cols                         # residual read by the epilogue (written back below)
        return w + act_in + 2 * B * out_cols

    import gc
    res, tot_b, tot_t, launches = {}, 0.0, 0.0, 0
    for kind in kinds + ("all",):
        gc.collect()   # a stale CUDAGraph must not be finalised during the capture below
        sel = kinds if kind == "all" else (kind,)

        def body():
            for layer, (kc, vc) in zip(model.layers, engine.kv_caches):
                for kd in sel:
                    launch(layer, kc, vc, kd)
        body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        n = len(model.layers) * len(sel) * reps
        t = a.elapsed_time(b) * 1e-3
        by = 0
        for layer in model.layers:
            for kd in sel:
                lin = {"qkv": layer.qkv_proj, "o": layer.o_proj, "gate_up": layer.gate_up, "down": layer.down_proj}[kd]
                by += nbytes(kd, lin)
        by *= reps
        if kind == "all":
            tot_b, tot_t, launches = by, t, n
        else:
            res[kind] = {"GB/s": round(by / t / 1e9, 1), "us": round(t / n * 1e6, 2)}
    return tot_b, tot_t, launches, res


def measure_verify_plans(model, eng, world, reps=10):
    """N > 1: the verify forward (W4A16, T = batch x (k+1)) under BOTH plans -- decoder layers replicated / sharded with
    three collectives per layer -- timed with events around a replayed hipGraph of that forward alone (eagerly where the
    communicator cannot be captured), MAX over ranks.  With `tp_collective_us` this is what replaces the planning
    constants: the plan is chosen from numbers of the job's own hardware."""
    import gc
    import torch.distributed as dist
    res, keep = {}, model.tp.shard_layers

    def fwd():
        model.forward(eng.v_tokens, eng.v_pos, eng.kv_caches, eng.md_verify, eng.scratch_verify, w4a4=False)
    for name, plan in (("replicated", False), ("sharded", True)):
        model.tp.shard_layers = plan
        fwd()
        torch.cuda.synchronize()
        fn, mode, g = fwd, "eager", None
        try:
            gc.collect()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fwd()
        except Exception:
            torch.cuda.synchronize()
            g = None
        # graph or eager must be ONE decision of all ranks (a capture records collectives, it issues none): a rank that
        # replays while another runs eagerly would issue mismatched collectives and hang the job
        from qspec_amd.parallel import agree_all
        if agree_all(model.tp, g is not None, eng.device):
            g.replay()
            fn, mode = g.replay, "graph"
        barrier(world)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        t = torch.tensor([a.elapsed_time(b) / reps], dtype=torch.float64,
                         device=eng.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        res[name] = {"ms": round(float(t.item()), 4), "mode": mode}
    model.tp.shard_layers = keep
    return res


PMC_FILES = ("r04_pmc_fetch_size.json", "r04_pmc_fetch_size_llama-3-8b_bs32_k5.json", "r04_pmc_fetch_size_llama-2-13b_bs4_k3.json",
             "r04_pmc_fetch_size_llama-3-70b_bs8_k3.json", "r04_pmc_fetch_size_tinyllama-1.1b_bs1_k3.json",
             "r03_pmc_fetch_size.json", "r03_pmc_fetch_size_llama-3-8b_bs32_k5.json", "r02_pmc_fetch_size.json")
PMC_SOURCE = "profiles/r0*_pmc_fetch_size*.json"


def pmc_traffic_per_launch(model_name, batch, k):
    """HBM read bytes per launch of the dominant kernel from the PMC pass committed under profiles/ (a separate
    `rocprofv3 --pmc FETCH_SIZE --kernel-trace` run of scripts/profile_cycle.py; FETCH_SIZE KiB x 1024 x 2 = the gfx950
    correction for wide streaming reads, MI355X_MICROARCH.md).  The summary is keyed by workload: a number is returned
    only when its (model, batch, k) match this run, else None."""
    d = None
    for name in PMC_FILES:           # the headline workload's pass, then the other committed workloads
        try:
            c = json.load(open(os.path.join(ROOT, "profiles", name)))
        except OSError:
            continue
        wl = c.get("workload", {})
        if (wl.get("model"), wl.get("batch"), wl.get("k")) == (model_name, batch, k):
            d = c
            break
    if d is None:
        return None
    tot = n = 0
    for name, v in d.get("kernels", {}).items():
        if "gemm_w4a4_stream_kernel" in name or "gemm_w4a4_longk_kernel" in name:
            tot += v["hbm_read_bytes_per_launch_corrected"] * v["launches"]
            n += v["launches"]
    return int(tot / n) if n else None


def cpu_baseline(model, args, rho):
    """The CPU oracle (port of the reference arithmetic) on this host, as BASELINE.md section 3 plans it: same k / batch,
    the same decode state as the GPU's timed region (prompt_len tokens of context per sequence: the KV cache is
    filled directly, the CPU prompt pass itself is not part of the sample), `cpu_warmup` untimed cycles, then the
    MEDIAN of `cpu_cycles` timed ones.  Bounded: `cpu_layers` full-width layers + lm_head; the layer share of a cycle
    is scaled to the full depth (the lm_head / sampler share is measured separately and not scaled)."""
    import copy
    import numpy as np
    import oracle as O
    from oracle.model import OracleEngine, OracleModel
    O.build()
    cfg = model.config
    L = min(args.cpu_layers, cfg.num_hidden_layers)
    om = OracleModel.from_torch_model(model, 16, max_layers=L)
    om.cfg = copy.copy(cfg)
    om.cfg.num_hidden_layers = L
    rng = np.random.default_rng(0)
    cores = os.cpu_count() or 1
    n_cyc = args.cpu_warmup + args.cpu_cycles
    ctx = args.prompt_len
    eng = OracleEngine(om, args.k, args.batch, ctx + (n_cyc + 1) * (args.k + 1) + 16, 16)
    eng.agreement_rho = rho
    for kc, vc in eng.kv:
        kc[...] = (rng.standard_normal(kc.shape) * 0.5).astype(np.float16)
        vc[...] = (rng.standard_normal(vc.shape) * 0.5).astype(np.float16)
    eng.seq_lens[:] = ctx + 1
    eng.last_token[:] = rng.integers(0, cfg.vocab_size, args.batch)
    V = cfg.vocab_size
    times, emitted = [], []
    for c in range(n_cyc):
        U = rng.random((args.batch, args.k)).astype(np.float32)
        E = rng.exponential(1.0, (args.batch, args.k, V)).astype(np.float32)
        n0 = sum(len(g) for g in eng.generated)
        t0 = time.perf_counter()
        eng.step(U, E)
        dt = time.perf_counter() - t0
        if c >= args.cpu_warmup:
            times.append(dt)
            emitted.append(sum(len(g) for g in eng.generated) - n0)
    dt = float(np.median(times))
    x = (rng.standard_normal((args.batch, cfg.hidden_size))).astype(np.float16)
    O.softmax_argmax(om.logits(x))
    t1 = time.perf_counter()
    O.softmax_argmax(om.logits(x))
    head = time.perf_counter() - t1
    head_cycle = head * (args.k + (args.k + 1))          # k draft heads at T=B, one verify head at T=B(k+1)
    layers_cycle = max(dt - head_cycle, 0.0)
    full = layers_cycle * cfg.num_hidden_layers / L + head_cycle
    return {"value": round(float(np.mean(emitted)) / full, 4), "unit": "tokens/s", "cores": cores, "kind": "port",
            "ms_per_step": round(full * 1e3, 1),
            "sample": f"median of {args.cpu_cycles} cycles after {args.cpu_warmup} warm-ups of the CPU oracle on {L}/{cfg.num_hidden_layers} "
                      f"full-width layers + lm_head, k={args.k} bs={args.batch}, {ctx}-token context per sequence (KV filled "
                      f"directly, prompt pass not sampled); layer share x{cfg.num_hidden_layers / L:g} extrapolated ({dt:.2f} s/cycle "
                      f"measured -> {full:.2f} s/cycle full depth), OpenMP {cores} threads"}


def timed_run(QSpecEngine, model, k, batch, prompts, prompt_len, agreement, warmup, steps, repeats, seed, world, dev):
    """W untimed cycles, then `repeats` timed regions of exactly `steps` cycles each, every region bracketed by a barrier +
    torch.cuda.synchronize() on both sides, MAX over ranks per region.  Returns the engine and one record per region:
    (seconds, accepted, emitted, draft tokens)."""
    total = warmup + steps + 2          # (every repeat restarts from the state behind the warm-up)
    eng = QSpecEngine(model, k, batch, max_model_len=prompt_len + total * (k + 1) + 32, block_size=16,
                      max_new_tokens=total * (k + 1) + 8, use_graph=True, seed=seed)
    eng.set_agreement(agreement)
    eng.add_sequences(prompts)               # prefill (W4A16), untimed: inputs are resident when timing starts
    for _ in range(warmup):
        eng.step()
    regions = []
    state0 = eng.save_region_state()
    for r in range(repeats):
        if r:
            eng.restore_region_state(state0)
        barrier(world)
        c0 = eng.sampler.counters.clone()
        barrier(world)
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.step()
        barrier(world)
        dt = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        acc, emit, draft = (int(v) for v in (eng.sampler.counters - c0).tolist())
        regions.append((dt, acc, emit, draft))
    assert eng.error_flag() == 0, "a device-side hand-off timed out"
    return eng, regions


def median_region(regions):
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    return regions[order[len(order) // 2]]


def roofline_block(model, eng, cfg, batch, k):
    tot_b, tot_t, n, per_shape = measure_dominant_kernel(model, eng)
    achieved = tot_b / tot_t / 1e9
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
            "frac": round(achieved / 8000.0, 4), "traffic": pmc_traffic_per_launch(cfg.name, batch, k),
            "traffic_source": PMC_SOURCE + " (builder's separate `rocprofv3 --pmc FETCH_SIZE` pass of the same "
                              "workload, committed; not collected in this run)",
            "kernel": "qspec::gemm_w4a4_stream_kernel (the four decoder GEMM launches of a draft forward, M = batch, in "
                      "the forms the cycle launches for this shape: LN+int4-quant prologue -> qkv+RoPE+KV-write / "
                      "gate_up+silu*up and o_proj / down_proj + residual add where built, the plain (xq, xs) forms otherwise)",
            "launches": n, "avg_launch_us": round(tot_t / n * 1e6, 2),
            "bytes_per_launch_avg": int(tot_b / n), "per_shape": per_shape}


def cycle_numbers(cfg, k, batch, sec_per_step):
    alg_bytes_cycle = (k + 1) * cfg.algorithmic_bytes_per_forward()
    macs_per_token = 2 * cfg.packed_weight_bytes_per_layer() * cfg.num_hidden_layers + cfg.vocab_size * cfg.hidden_size
    flops_cycle = 2.0 * macs_per_token * batch * (2 * k + 1)
    return {"cycle_hbm_GBps_algorithmic": round(alg_bytes_cycle / sec_per_step / 1e9, 1),
            "cycle_hbm_frac_of_8TBps": round(alg_bytes_cycle / sec_per_step / 8e12, 4),
            # matrix-core work of a cycle: 2 flop per weight per token (layers + lm_head), k x B draft + (k+1) x B verify
            # tokens, against the 2.5 PFLOP/s dense fp16 / int8-as-fp16-equivalent peak
            "cycle_mfma_TFLOPs": round(flops_cycle / sec_per_step / 1e12, 2),
            "cycle_mfma_frac_of_2.5PF": round(flops_cycle / sec_per_step / 2.5e15, 4)}


def measure_other_config(QSpecEngine, name, k, batch, label, args, rho, dev):
    """One of the other BASELINE.json configs on this GPU: same engine code, same synthetic-agreement knob, same timing
    brackets as the headline (warm-up, `repeats` regions of `other_steps` cycles, median), plus the dominant kernel's
    roofline for that shape.  The model is built, measured and freed inside."""
    import gc
    from qspec_amd.model import CONFIGS, QuarotLlamaForCausalLM
    cfg = CONFIGS[name]
    t_build = time.perf_counter()
    model = QuarotLlamaForCausalLM(cfg, dev).init_synthetic(args.seed, args.lm_head_std)
    g = torch.Generator().manual_seed(args.seed)
    prompt_len = min(args.prompt_len, cfg.max_position_embeddings // 2)
    prompts = [torch.randint(0, cfg.vocab_size, (prompt_len,), generator=g).tolist() for _ in range(batch)]
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    eng, regions = timed_run(QSpecEngine, model, k, batch, prompts, prompt_len, rho, min(args.warmup, 5), args.other_steps,
                             args.repeats, args.seed, 1, dev)
    dt, acc, emit, draft = median_region(regions)
    sec = dt / args.other_steps
    out = {"workload": f"{label}; prompt_len={prompt_len} greedy, synthetic int4 weights, synthetic agreement {rho}",
           "model": name, "num_speculative_tokens": k, "batch": batch,
           "value": round(emit / dt, 2), "unit": "tokens/s", "steps": args.other_steps, "ms_per_step": round(sec * 1e3, 4),
           "ms_per_step_repeats": [round(r[0] / args.other_steps * 1e3, 4) for r in regions],
           "draft_acceptance_rate": round(acc / draft, 4) if draft else None,
           "system_efficiency": round(emit / ((draft // k) * (k + 1)), 4) if draft else None,
           "capture": "graph" if eng._graph is not None else "eager", "build_s": round(t_build, 1)}
    out.update(cycle_numbers(cfg, k, batch, sec))
    if not args.no_roofline:
        out["roofline"] = roofline_block(model, eng, cfg, batch, k)
    del eng, model
    gc.collect()
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank, world, local = dist_setup(args.gpus)
    dev = f"cuda:{local}"
    torch.cuda.set_device(dev)
    from qspec_amd.model import CONFIGS, QuarotLlamaForCausalLM
    QSpecEngine = make_bench_engine_class()
    cfg = CONFIGS[args.model]
    if world > 1:
        from qspec_amd import parallel
        # the plan of the verify pass (shard the decoder layers or not) comes from the collectives' cost MEASURED here
        model = parallel.build_tp_model(cfg, dev, world, rank, seed=args.seed, lm_head_std=args.lm_head_std,
                                        tokens=args.batch * (args.k + 1), draft_tokens=args.batch)
    else:
        model = QuarotLlamaForCausalLM(cfg, dev).init_synthetic(args.seed, args.lm_head_std)
    rho = None if args.agreement.lower() == "none" else float(args.agreement)
    g = torch.Generator().manual_seed(args.seed)
    prompts = [torch.randint(0, cfg.vocab_size, (args.prompt_len,), generator=g).tolist() for _ in range(args.batch)]

    def run(agreement, warmup, steps, repeats=1):
        return timed_run(QSpecEngine, model, args.k, args.batch, prompts, args.prompt_len, agreement, warmup, steps, repeats,
                         args.seed, world, dev)

    def run_e2e(agreement, max_tokens):
        """demo.py:139-160: prompt pass + decode until every request has max_tokens tokens; tokens / wall time.  The
        host reads the batch's emitted counts once per cycle, as the worker does."""
        n_cyc = max_tokens + 4            # worst case: one token per cycle
        eng = QSpecEngine(model, args.k, args.batch, max_model_len=args.prompt_len + n_cyc * (args.k + 1) + 32,
                          block_size=16, max_new_tokens=max_tokens + 2 * (args.k + 1) + 8, use_graph=True, seed=args.seed)
        eng.set_agreement(agreement)
        eng.add_sequences(prompts); eng.step(); eng.sync_lens()     # capture outside the timed region ...
        for b in range(args.batch):
            eng.free_slot(b)                                        # ... then start over from empty slots
        barrier(world)
        t0 = time.perf_counter()
        eng.add_sequences(prompts)
        t_prefill = time.perf_counter() - t0
        cycles, done = 0, {}
        while len(done) < args.batch:
            eng.step()
            cycles += 1
            eng.sync_lens()                                         # the per-cycle host read (as the worker's)
            lens = list(eng._gen_ub)
            for b, n in enumerate(lens):                            # a request at max_tokens leaves the batch
                if b not in done and n >= max_tokens:
                    done[b] = max_tokens
                    eng.free_slot(b)
        barrier(world)
        dt = time.perf_counter() - t0
        total = sum(done.values())
        del eng
        import gc
        gc.collect()
        return total / dt, t_prefill, cycles, dt

    eng, regions = run(rho, args.warmup, args.steps, max(1, args.repeats))
    dt, acc, emit, draft = median_region(regions)     # ms_per_step / value: the MEDIAN of the timed regions
    per_step = [r[0] / args.steps * 1e3 for r in regions]
    rate = acc / draft if draft else float("nan")
    eff = emit / ((draft // args.k) * (args.k + 1)) if draft else float("nan")
    agree_txt = "weights' own agreement" if rho is None else f"synthetic draft/target agreement {rho}"
    out = {
        "metric": "accepted_tokens_per_s" if rho is None else f"accepted_tokens_per_s_at_synthetic_agreement_{rho:g}",
        "value": round(emit / dt, 2), "unit": "tokens/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "w4a4: s4 x s4 -> i32 (i8 MFMA); w4a16: f16 x s4 -> f32 (f16 MFMA)", "data": "synthetic",
        "config": {"workload": f"{cfg.name} QSpec W4A4-draft/W4A16-verify k={args.k} bs={args.batch} TP={world} "
                               f"prompt_len={args.prompt_len} greedy, synthetic int4 weights (SURVEY 8d), {agree_txt}; "
                               f"1 step = 1 cycle = {args.k} draft fwd + 1 verify fwd + rejection sampling",
                   "num_speculative_tokens": args.k, "batch": args.batch, "parallelism": f"tp{world}",
                   "tp_plan": (None if world == 1 else
                               ("verify pass: o_proj/gate_up/down_proj sharded (3 collectives per layer) + vocab-parallel lm_head"
                                if model.tp.shard_layers else
                                "verify pass: vocab-parallel lm_head + all-gather; decoder layers replicated (their collectives "
                                "would cost more than the weight stream they save at this size); draft pass replicated")),
                   "tp_draft_pass": (None if world == 1 else
                                     ("decoder layers replicated (zero collectives on activations); lm_head vocab-parallel + all-gather "
                                      "of the fp16 logit slices" if model.tp.shard_draft_vocab else
                                      "replicated, zero collectives (the logits all-gather would cost more than the lm_head stream it saves)")),
                   "tp_draft_vocab_gather_us": (None if world == 1 else getattr(model.tp, "vocab_gather_us", None)),
                   "tp_collective": (None if world == 1 else model.tp.backend),
                   "tp_plan_basis": (None if world == 1 else getattr(model.tp, "plan_basis", None)),
                   "tp_collective_us": (None if world == 1 else getattr(model.tp, "collective_us", None)),
                   "agreement": rho,
                   # what THIS run did: "graph" = the whole cycle incl. its collectives was captured and replayed (one rank
                   # per GPU: the only configuration in which an RCCL-in-graph claim means anything)
                   "capture": ("graph" if eng._graph is not None else
                               ("draft-graph+eager-verify" if eng._graph_draft is not None else "eager"))},
        # the --steps region timed `repeats` times back to back (same engine, same graph): ms_per_step / value above are the
        # median region's; spread = (max - min) / median
        "repeats": len(regions), "ms_per_step_repeats": [round(v, 4) for v in per_step],
        "ms_per_step_spread": round((max(per_step) - min(per_step)) / sorted(per_step)[len(per_step) // 2], 4),
        "draft_acceptance_rate": round(rate, 4), "system_efficiency": round(eff, 4),
        "accepted_tokens": acc, "emitted_tokens": emit, "draft_tokens": draft,
    }
    out.update(cycle_numbers(cfg, args.k, args.batch, dt / args.steps))
    if rho is not None and args.natural_steps > 0:
        # the same engine code with the random weights' own agreement (acceptance ~ a few %): reported beside the headline
        _, reg2 = run(None, min(args.warmup, 5), args.natural_steps)
        dt2, acc2, emit2, draft2 = reg2[0]
        out["natural_agreement"] = {"value": round(emit2 / dt2, 2), "unit": "tokens/s", "steps": args.natural_steps,
                                    "ms_per_step": round(dt2 / args.natural_steps * 1e3, 4),
                                    "draft_acceptance_rate": round(acc2 / draft2, 4) if draft2 else None,
                                    "system_efficiency": round(emit2 / ((draft2 // args.k) * (args.k + 1)), 4) if draft2 else None}
    if args.e2e_max_tokens > 0:
        tps, t_pre, cycles, wall = run_e2e(rho, args.e2e_max_tokens)
        out["e2e_incl_prefill"] = {"value": round(tps, 2), "unit": "tokens/s", "max_tokens": args.e2e_max_tokens,
                                   "prompt_tokens": args.batch * args.prompt_len, "prefill_ms": round(t_pre * 1e3, 2),
                                   "cycles": cycles, "wall_s": round(wall, 4),
                                   "what": "demo.py:139-160: generated tokens / (prompt pass + decode) wall time, requests "
                                           "leave the batch at max_tokens"}
    if world > 1:
        out["config"]["tp_verify_forward"] = measure_verify_plans(model, eng, world)
    if rank == 0 and not args.no_roofline:   # the draft pass is replicated under TP: rank 0's launches are every rank's
        out["roofline"] = roofline_block(model, eng, cfg, args.batch, args.k)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(model, args, rho)
    # ---- the other BASELINE.json configs on this GPU (N = 1, default workload only: the driver's run)
    others = []
    if args.other_configs == "auto":
        if world == 1 and (args.model, args.k, args.batch) == ("llama-3-8b", 3, 4):
            others = list(OTHER_CONFIGS)
    elif args.other_configs not in ("none", ""):
        for item in args.other_configs.split(","):
            nm, kk, bb = item.split(":")
            others.append((nm, int(kk), int(bb), f"{nm} k={kk} bs={bb}"))
    if others and world == 1:
        import gc
        del eng, model
        gc.collect()
        torch.cuda.empty_cache()
        out["other_configs"] = {}
        for nm, kk, bb, label in others:
            try:
                out["other_configs"][f"{nm}_k{kk}_bs{bb}"] = measure_other_config(QSpecEngine, nm, kk, bb, label, args, rho, dev)
            except Exception as exc:   # the headline line must survive a failure here: reported, not hidden
                out["other_configs"][f"{nm}_k{kk}_bs{bb}"] = {"error": repr(exc)[:300]}
                torch.cuda.synchronize()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        barrier(world)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
