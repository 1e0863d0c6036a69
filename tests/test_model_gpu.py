"""End-to-end parity on a tiny QSpec model: fused HIP forward vs module-wise HIP forward (bit-exact) and vs the
CPU oracle model (tolerance where attention / W4A16 accumulate in an order only the hardware fixes), then the
whole draft->verify->accept cycle against the oracle engine with injected random draws."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def tiny_cfg():
    from qspec_amd.model import QuarotLlamaConfig
    # 8 heads x 128, GQA group 4, I = 28 * 128 (had28 (x) H128), 2 layers
    return QuarotLlamaConfig(1024, 3584, 8, 2, 2, 2048, 1e-5, 10000.0, 512, "tiny")


@pytest.fixture(scope="module")
def tiny():
    assert torch.cuda.is_available()
    from qspec_amd.model import QuarotLlamaForCausalLM
    return QuarotLlamaForCausalLM(tiny_cfg(), DEV).init_synthetic(seed=1, lm_head_std=0.05)


def make_inputs(model, rng, ctx_lens, q_len, block_size=16):
    """Random paged KV history + q_len new tokens per sequence."""
    from qspec_amd.model import AttentionMetadata
    cfg = model.config
    B = len(ctx_lens)
    max_blocks = max((c + block_size - 1) // block_size for c in ctx_lens) + 1
    nb = B * max_blocks
    bt = np.arange(nb, dtype=np.int32).reshape(B, max_blocks)
    shape = (nb, block_size, cfg.num_key_value_heads, cfg.head_dim)
    kv_np = [((rng.standard_normal(shape) * 0.5).astype(np.float16), (rng.standard_normal(shape) * 0.5).astype(np.float16))
             for _ in range(cfg.num_hidden_layers)]
    T = B * q_len
    ids = rng.integers(0, cfg.vocab_size, T).astype(np.int64)
    pos = np.concatenate([np.arange(c - q_len, c) for c in ctx_lens]).astype(np.int64)
    slots = np.concatenate([bt[b, (np.arange(c - q_len, c)) // block_size].astype(np.int64) * block_size
                            + np.arange(c - q_len, c) % block_size for b, c in enumerate(ctx_lens)])
    q_start = (np.arange(B + 1) * q_len).astype(np.int32)
    ctx = np.array(ctx_lens, np.int32)
    n_splits = (max(ctx_lens) + 127) // 128
    t = lambda a: torch.from_numpy(a).to(DEV)  # noqa: E731
    md = AttentionMetadata(t(slots), t(bt), t(ctx), t(q_start), q_len, n_splits)
    kv_t = [(t(k), t(v)) for k, v in kv_np]
    return dict(ids=ids, pos=pos, slots=slots, bt=bt, ctx=ctx, q_start=q_start, kv_np=kv_np, kv_t=kv_t, md=md, T=T,
                n_splits=n_splits, ids_t=t(ids), pos_t=t(pos))


@pytest.mark.parametrize("w4a4,ctx_lens,q_len", [(True, [40, 130, 7, 260], 1), (False, [40, 130, 9, 260], 4)])
def test_fused_forward_equals_modulewise(tiny, w4a4, ctx_lens, q_len):
    """The fused layer (7 / 9 launches) must reproduce the reference-order, one-op-per-module path bit for bit."""
    from qspec_amd.model import Scratch
    rng = np.random.default_rng(0)
    inp = make_inputs(tiny, rng, ctx_lens, q_len)
    s = Scratch(tiny.config, inp["T"], len(ctx_lens), q_len, inp["n_splits"], DEV)
    kv_a = [(k.clone(), v.clone()) for k, v in inp["kv_t"]]
    kv_b = [(k.clone(), v.clone()) for k, v in inp["kv_t"]]
    a = tiny.forward(inp["ids_t"], inp["pos_t"], kv_a, inp["md"], s, w4a4=w4a4).clone()
    b = tiny.forward_modulewise(inp["ids_t"], inp["pos_t"], kv_b, inp["md"], w4a4=w4a4)
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    for (ka, va), (kb, vb) in zip(kv_a, kv_b):
        assert torch.equal(ka, kb) and torch.equal(va, vb)


@pytest.mark.parametrize("w4a4,ctx_lens,q_len", [(True, [40, 130, 7, 260], 1), (False, [40, 130, 9, 260], 4)])
def test_forward_matches_oracle_model(tiny, oracle, w4a4, ctx_lens, q_len):
    from oracle.model import OracleModel
    from qspec_amd.model import Scratch
    rng = np.random.default_rng(1)
    inp = make_inputs(tiny, rng, ctx_lens, q_len)
    om = OracleModel.from_torch_model(tiny, 16)
    kv_np = [(k.copy(), v.copy()) for k, v in inp["kv_np"]]
    ref = om.forward(inp["ids"], inp["pos"], kv_np, inp["slots"], inp["bt"], inp["ctx"], inp["q_start"], w4a4)
    s = Scratch(tiny.config, inp["T"], len(ctx_lens), q_len, inp["n_splits"], DEV)
    out = tiny.forward(inp["ids_t"], inp["pos_t"], inp["kv_t"], inp["md"], s, w4a4=w4a4)
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.float64)
    # layer-norm output is O(1); int4 paths amplify a rare 1-ulp attention difference into one quantisation
    # step somewhere, so the bar is statistical for W4A4 and 1e-3-class for W4A16
    diff = np.abs(got - ref.astype(np.float64))
    if w4a4:
        # chaotic by nature: one different int4 (e.g. the abs-max element of a row) re-scales a whole row downstream
        assert np.median(diff) < 1e-3 and np.quantile(diff, 0.99) < 0.1, (np.median(diff), np.quantile(diff, 0.99), diff.max())
    else:
        # W4A16: against the noise floor of the comparison -- the oracle vs itself with fp32-accumulating W4A16 GEMMs (a second
        # admissible implementation; see test_full_depth_llama3_8b_verify_forward_and_cycle) -- instead of a fixed 2e-2
        floor, rel = _w4a16_noise_floor(om, inp, ref), _rel3(got, ref)
        print(f"w4a16 forward, err / 1e-3: max {rel.max():.2f} q99 {np.quantile(rel, 0.99):.2f}; floor max {floor.max():.2f} q99 {np.quantile(floor, 0.99):.2f}")
        assert np.quantile(rel, 0.99) < max(1.5 * np.quantile(floor, 0.99), 2.0) and rel.max() < max(2.0 * floor.max(), 4.0), \
            (np.quantile(rel, 0.99), np.quantile(floor, 0.99), rel.max(), floor.max())
    # layer-0 KV written by the HIP path equals the oracle's exactly for W4A4 (no attention upstream of it)
    if w4a4:
        k_hip = inp["kv_t"][0][0].cpu().numpy()
        assert np.array_equal(k_hip.view(np.uint16), kv_np[0][0].view(np.uint16))
    logits = tiny.compute_logits(out, s).cpu().numpy().astype(np.float64)
    ref_logits = om.logits(ref).astype(np.float64)
    assert np.quantile(np.abs(logits - ref_logits), 0.99) < (0.1 if w4a4 else 3e-2)


def _w4a16_noise_floor(om, inp, ref):
    """|oracle(fp32-accumulate W4A16) - oracle(fp64-accumulate)| on the same inputs: how far two implementations that
    both meet the per-stage 1e-3 bar are apart at the end of this model's forward."""
    om.w4a16_f32acc = True
    try:
        kv2 = [(k.copy(), v.copy()) for k, v in inp["kv_np"]]
        ref2 = om.forward(inp["ids"], inp["pos"], kv2, inp["slots"], inp["bt"], inp["ctx"], inp["q_start"], False)
    finally:
        om.w4a16_f32acc = False
    # The floor moves ONE rounding source (the W4A16 accumulate order); the other fp16 rounding points (norm, Hadamard, attention
    # output) are covered by the fixed minimum of the bars: 2 units at the 99% quantile, 4 units (~4 fp16 ulps) at the max.
    return _rel3(ref2, ref)


def _rel3(got, ref):
    """|got - ref| in units of 1e-3 x max(1, |ref|): north_star's tolerance, relative above 1 (an fp16 ulp at |x| = 8 is 8e-3)."""
    got, ref = np.asarray(got).astype(np.float64), np.asarray(ref).astype(np.float64)
    return np.abs(got - ref) / (1e-3 * np.maximum(1.0, np.abs(ref)))


def _engine_cycle_check(model, oracle, k, B, prompt_lens, cycles, seed, tv_bars, sync_kv=False, d_q98=None):
    """Three full cycles (k=3, B=4).  Two HIP-vs-oracle differences are legitimate: fp32 summation order inside
    attention / the W4A16 MFMA, and (a consequence) a rare different int4 value downstream.  So:
      * numerics: the oracle engine is teacher-forced with the GPU's draft tokens; its draft / target
        distributions must be close to the GPU's (total-variation distance per row);
      * logic: everything discrete -- accept mask, recovered ids, output layout, counters, sequence state,
        KV slot bookkeeping -- must be EXACT given the GPU's own distributions and the injected draws."""
    from oracle.model import OracleEngine, OracleModel
    from qspec_amd.spec_decode import QSpecEngine
    tiny = model
    rng = np.random.default_rng(seed)
    V = tiny.config.vocab_size
    prompts = [rng.integers(0, V, n).tolist() for n in prompt_lens]
    eng = QSpecEngine(tiny, k, B, max_model_len=256, block_size=16, max_new_tokens=64, use_graph=False, seed=0)
    eng.add_sequences(prompts)
    oe = OracleEngine(OracleModel.from_torch_model(tiny, 16), k, B, 256, 16)
    oe.add_sequences(prompts)
    # prefill: greedy target token.  The W4A16 prompt pass is 1e-3-close, not bit-identical, so a near-tie of the two
    # best logits may resolve differently: then the oracle must rate the GPU's token within 2 % of its own best, and
    # it continues from the GPU's token (teacher forcing, as for the draft tokens below)
    for b, (tg, to) in enumerate(zip(eng.last_token.tolist(), oe.last_token.tolist())):
        if tg != to:
            pr = oe.prefill_probs[b]
            assert pr[tg] > 0.98 * pr[to], (b, tg, to, pr[tg], pr[to])
            oe.last_token[b] = tg
            oe.generated[b][-1] = tg
    gen = [[int(t)] for t in eng.last_token.tolist()]
    counters = np.zeros(3, np.int64)
    all_tv_d, all_tv_t = [], []
    for cyc in range(cycles):
        U = rng.random((B, k)).astype(np.float32)
        E = rng.exponential(1.0, (B, k, V)).astype(np.float32)
        eng.inject_uniform, eng.inject_exponential = torch.from_numpy(U).to(DEV), torch.from_numpy(E).to(DEV)
        L0 = eng.seq_lens.cpu().numpy().copy()
        if sync_kv:   # teacher-force the history too: the oracle continues from the GPU's KV cache (written by the
            # 1e-3-close W4A16 passes), so that the draft pass is compared on IDENTICAL inputs
            for (kg, vg), (ko, vo) in zip(eng.kv_caches, oe.kv):
                ko[...] = kg.cpu().numpy()
                vo[...] = vg.cpu().numpy()
        eng.step()
        torch.cuda.synchronize()
        d_ids = eng.draft_ids_kb.t().cpu().numpy()
        d_probs = eng.draft_probs_kbv.transpose(0, 1).cpu().numpy()
        t_probs, t_toks = eng.target_probs.cpu().numpy(), eng.target_tokens.cpu().numpy()
        out = eng.out_tokens.cpu().numpy()
        # ---- numerics (teacher forced)
        r = oe.step(U, E, forced_draft_ids=d_ids, forced_out=out)
        tv_d = 0.5 * np.abs(r["draft_probs"] - d_probs).sum(-1)
        tv_t = 0.5 * np.abs(r["target_probs"] - t_probs).sum(-1)
        all_tv_d.append(tv_d)
        all_tv_t.append(tv_t)
        # ---- logic (exact)
        o_out, o_acc, o_rec, c = oracle.rejection_sample(t_probs, t_toks[:, k], d_probs, d_ids, U, E)
        assert np.array_equal(out, o_out), cyc
        assert np.array_equal(eng.accepted.cpu().numpy().astype(bool), o_acc), cyc
        rej = ~o_acc
        assert np.array_equal(eng.recovered.cpu().numpy()[rej], o_rec[rej]), cyc
        counters += np.array(c)
        n_emit = (out != -1).sum(1)
        assert (n_emit >= 1).all() and np.array_equal(eng.seq_lens.cpu().numpy(), L0 + n_emit)
        for b in range(B):
            em = [int(t) for t in out[b] if t != -1]
            assert out[b, :len(em)].tolist() == em            # -1 only as a suffix
            gen[b].extend(em)
            assert int(eng.last_token[b]) == em[-1]
        assert eng.seq_lens.tolist() == oe.seq_lens.tolist() and eng.last_token.tolist() == oe.last_token.tolist()
        # verify inputs were assembled as [last, d_1..d_k] at positions L-1..L-1+k over the sequence's own blocks
        vt = eng.v_tokens.view(B, k + 1).cpu().numpy()
        assert np.array_equal(vt[:, 1:], d_ids)
        assert np.array_equal(eng.v_pos.view(B, k + 1).cpu().numpy(), (L0 - 1)[:, None] + np.arange(k + 1)[None])
        bt = eng.block_tables.cpu().numpy()
        pos = eng.v_pos.view(B, k + 1).cpu().numpy()
        exp_slots = np.take_along_axis(bt, (pos // 16).astype(np.int64), 1).astype(np.int64) * 16 + pos % 16
        assert np.array_equal(eng.v_slots.view(B, k + 1).cpu().numpy(), exp_slots)
    # W4A4 is chaotic (a single different int4 can re-scale a row), so the draft distributions are compared
    # statistically: most of them agree to fp32 rounding, none is far off; the W4A16 target must stay close always
    by_step = np.concatenate(all_tv_d).reshape(-1, k).max(0)      # step 0 runs on identical inputs under sync_kv
    print("TV draft max by draft step:", " ".join(f"{v:.2e}" for v in by_step))
    tv_d, tv_t = np.concatenate(all_tv_d).ravel(), np.concatenate(all_tv_t).ravel()
    # (measured: a 1e-7 attention difference flips an fp16 ulp in ~20% of rows, the int4 pipeline amplifies it to
    # a total-variation distance of ~0.05-0.08 after two layers)
    d_med, d_max, t_med, t_max = tv_bars
    print(f"TV draft median {np.median(tv_d):.4f} max {tv_d.max():.4f} min {tv_d.min():.2e}; "
          f"target median {np.median(tv_t):.5f} max {tv_t.max():.4f}")
    assert np.median(tv_d) < d_med and tv_d.max() < d_max and tv_d.min() < 1e-3, tv_d
    if d_q98 is not None:
        print(f"TV draft rows above {d_q98:g}: {int((tv_d >= d_q98).sum())} of {tv_d.size}")
        assert np.quantile(tv_d, 0.98) < d_q98, np.sort(tv_d)[-8:]
    assert np.median(tv_t) < t_med and tv_t.max() < t_max, tv_t
    assert eng.generated() == gen
    m = eng.metrics()
    assert [m.accepted_tokens, m.emitted_tokens, m.draft_tokens] == counters.tolist()
    rate, eff = oracle.spec_metrics(*counters.tolist(), k)
    assert abs(m.draft_acceptance_rate - rate) < 1e-12 and abs(m.system_efficiency - eff) < 1e-12



def test_engine_cycle_matches_oracle_engine(tiny, oracle):
    _engine_cycle_check(tiny, oracle, 3, 4, (17, 33, 64, 5), 3, 2, (0.12, 0.6, 5e-3, 0.15))


# With the KV history teacher-forced as well (sync_kv) the draft pass starts every cycle on IDENTICAL inputs: its
# distributions are compared at what is measured, not at the chaotic-drift bars: median total variation < 1e-4 (measured
# 0.0000), 98 % of the rows < 1e-2 (measured: 0 rows above at bs = 4 and bs = 1, 1 of 320 at bs = 32 / k = 5).  The
# maximum cannot be bounded that tightly: draft steps 2..k attend to K/V the SAME cycle's earlier draft steps wrote, a
# last-bit attention difference there (hardware v_exp_f32 vs the oracle's qexpf) can flip one int4 and the row-absmax
# quantiser re-scales that row (measured once: 0.22).  A regression that moved every draft distribution by even 1 %
# fails the median and the quantile.  Target (W4A16) distributions: measured median 0.5-1.5e-3, max 4e-3.
SAME_HISTORY_BARS = (1e-4, 0.6, 3e-3, 2e-2)


def test_engine_cycle_matches_oracle_engine_same_history(tiny, oracle):
    _engine_cycle_check(tiny, oracle, 3, 4, (17, 33, 64, 5), 3, 2, SAME_HISTORY_BARS, sync_kv=True, d_q98=1e-2)


def test_engine_cycle_tinyllama_k3_bs1(oracle):
    """BASELINE.json configs[0] -- TinyLlama-1.1B, k = 3, bs = 1 -- through the ENGINE on the HIP path at that model's
    layer width (H = 2048, I = 5632 = had44 (x) H128, 32 heads / 4 kv heads of head_dim 64; 2 layers, small vocabulary):
    the dispatch that differs from the headline config -- generic head-size attention kernel without context split,
    un-fused QKV epilogue (rope_kv_write), had44 MLP transform, one sequence (B * kv heads = 4 groups) -- with the
    same exact-logic assertions (accept masks, recovered ids, layout, counters, KV slot bookkeeping) and the
    same-history bars on the distributions; then the captured hipGraph of that cycle against the eager cycle."""
    from qspec_amd.model import QuarotLlamaConfig, QuarotLlamaForCausalLM
    from qspec_amd.spec_decode import QSpecEngine
    cfg = QuarotLlamaConfig(2048, 5632, 32, 4, 2, 2048, 1e-5, 10000.0, 512, "tinyllama-1.1b-2layer")
    model = QuarotLlamaForCausalLM(cfg, DEV).init_synthetic(seed=7, lm_head_std=0.05)
    assert cfg.head_dim == 64
    _engine_cycle_check(model, oracle, 3, 1, (23,), 3, 21, SAME_HISTORY_BARS, sync_kv=True, d_q98=1e-2)
    outs = []
    prompt = np.random.default_rng(5).integers(0, cfg.vocab_size, 37).tolist()
    for use_graph in (False, True):
        eng = QSpecEngine(model, 3, 1, max_model_len=256, block_size=16, max_new_tokens=64, use_graph=use_graph, seed=9)
        eng.add_sequences([prompt])
        for _ in range(5):
            eng.step()
        outs.append((eng.generated(), eng.metrics()))
    assert outs[0] == outs[1] and len(outs[0][0][0]) >= 6


def test_engine_cycle_k5_bs32_llama3_8b_width(oracle):
    """BASELINE.json configs[2] (k = 5, bs = 32) through the ENGINE at the full Llama-3-8B layer width (2 layers, small
    vocabulary): the M > 16 dispatch -- separate norm launches, two-token-tile W4A4 streaming GEMMs, the per-wave
    attention kernel, the M-tiled W4A16 GEMMs at T = 192 -- with the same exact-logic assertions as the k = 3 / bs = 4
    test (accept masks, recovered ids, layout, counters, KV slot bookkeeping exact given the GPU's own distributions)."""
    from qspec_amd.model import QuarotLlamaConfig, QuarotLlamaForCausalLM
    cfg = QuarotLlamaConfig(4096, 14336, 32, 8, 2, 2048, 1e-5, 500000.0, 512, "llama-3-8b-2layer")
    model = QuarotLlamaForCausalLM(cfg, DEV).init_synthetic(seed=3, lm_head_std=0.05)
    lens = [5 + (7 * i) % 29 for i in range(32)]
    # with the history teacher-forced as well, the draft pass is (nearly always) bit-identical to the oracle
    _engine_cycle_check(model, oracle, 5, 32, lens, 2, 12, SAME_HISTORY_BARS, sync_kv=True, d_q98=1e-2)


def test_full_depth_llama3_8b_verify_forward_and_cycle(oracle):
    """The headline config at FULL depth and vocabulary (32 layers, H = 4096, I = 14336, V = 128256; BASELINE.json
    configs[1]) on the HIP path against the CPU oracle -- the widest comparison elsewhere is 3 layers / V = 1024.
      1. one verify forward (W4A16, T = 16 = 4 sequences x (k+1)) over a random paged KV history: the error of the
         final normed hidden state and of the fp16 logits is printed against north_star's 1e-3 and bounded;
      2. one k = 3 / bs = 4 cycle through the engine, teacher-forced (draft tokens, KV history), with injected draws:
         accept masks, recovered ids, output layout, counters and slot bookkeeping EXACT given the GPU's distributions;
         draft / target distributions within the same-history bars.
    The reference holds nothing at this size either (parity unpinned); the oracle is its restatement."""
    from oracle.model import OracleModel
    from qspec_amd.model import CONFIGS, QuarotLlamaForCausalLM, Scratch
    cfg = CONFIGS["llama-3-8b"]
    assert (cfg.num_hidden_layers, cfg.vocab_size, cfg.hidden_size) == (32, 128256, 4096)
    model = QuarotLlamaForCausalLM(cfg, DEV).init_synthetic(seed=0)
    rng = np.random.default_rng(8)
    ctx_lens, q_len = [40, 130, 9, 260], 4
    inp = make_inputs(model, rng, ctx_lens, q_len)
    om = OracleModel.from_torch_model(model, 16)
    kv_np = [(k.copy(), v.copy()) for k, v in inp["kv_np"]]
    ref = om.forward(inp["ids"], inp["pos"], kv_np, inp["slots"], inp["bt"], inp["ctx"], inp["q_start"], False)
    s = Scratch(cfg, inp["T"], len(ctx_lens), q_len, inp["n_splits"], DEV)
    out = model.forward(inp["ids_t"], inp["pos_t"], inp["kv_t"], inp["md"], s, w4a4=False)
    logits = model.compute_logits(out, s)
    torch.cuda.synchronize()
    ref_logits = om.logits(ref).astype(np.float64)

    def dist(what, got, ref):
        r = np.abs(got.astype(np.float64) - ref.astype(np.float64)) / (1e-3 * np.maximum(1.0, np.abs(ref.astype(np.float64))))
        q = np.quantile(r, [0.5, 0.9, 0.99, 0.999])
        print(f"full depth, verify T=16, {what}: err / 1e-3  median {q[0]:.3f}  q90 {q[1]:.3f}  q99 {q[2]:.3f}  "
              f"q99.9 {q[3]:.3f}  max {r.max():.3f};  within 1e-3: {(r <= 1).mean():.4f}")
        return r
    r_h = dist("HIP vs oracle, normed hidden", out.cpu().numpy(), ref)
    r_l = dist("HIP vs oracle, logits", logits.cpu().numpy(), ref_logits)
    # The noise floor of the comparison: the oracle against ITSELF with its W4A16 GEMMs replaced by a second admissible
    # implementation (fp32 accumulate in interleaved partial sums; every stage of it meets the 1e-3 bar against the fp64
    # one, tests/test_oracle_golden.py).  32 layers of fp16 roundings on a residual stream whose ulp is ~1e-3 of its
    # row norm carry two such implementations this far apart; the HIP path must not be further from the oracle than that
    # by more than a factor 1.5 (measured on the box: HIP 3.8 / 4.4 x 1e-3 median).
    om.w4a16_f32acc = True
    kv_np2 = [(k.copy(), v.copy()) for k, v in inp["kv_np"]]
    ref2 = om.forward(inp["ids"], inp["pos"], kv_np2, inp["slots"], inp["bt"], inp["ctx"], inp["q_start"], False)
    om.w4a16_f32acc = False
    n_h = dist("oracle(f32 acc) vs oracle, normed hidden", ref2, ref)
    n_l = dist("oracle(f32 acc) vs oracle, logits", om.logits(ref2), ref_logits)
    assert np.median(r_h) < 1.5 * np.median(n_h) and np.quantile(r_h, 0.99) < 1.5 * np.quantile(n_h, 0.99), \
        (np.median(r_h), np.median(n_h), np.quantile(r_h, 0.99), np.quantile(n_h, 0.99))
    assert np.median(r_l) < 1.5 * np.median(n_l) and np.quantile(r_l, 0.99) < 1.5 * np.quantile(n_l, 0.99), \
        (np.median(r_l), np.median(n_l), np.quantile(r_l, 0.99), np.quantile(n_l, 0.99))
    assert np.median(r_l) < 10.0 and r_l.max() < 100.0      # and an absolute ceiling: logits within 1e-2 / 1e-1
    # the greedy token agrees wherever the oracle's two best logits are not a near tie
    top2 = np.sort(ref_logits, -1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 0.05
    assert np.array_equal(logits.float().argmax(-1).cpu().numpy()[clear], ref_logits.argmax(-1)[clear])
    del s, out, logits, inp, kv_np, kv_np2
    # 32 layers deep the chaotic rows of draft steps 2..k are more frequent (measured: 3 of 12 rows above 1e-2, median
    # still 0.0000) and the target distributions carry the verify pass's 32-layer drift (measured median 3.6e-3 =
    # the noise floor above): median bar as everywhere, no quantile bar, target bars 8e-3 / 2e-2
    _engine_cycle_check(model, oracle, 3, 4, (9, 12, 6, 5), 1, 40, (1e-4, 0.6, 8e-3, 2e-2), sync_kv=True)

def _true_math_verify_forward(model, inp):
    """The verify forward in UN-ROUNDED math: fp64 throughout (torch on the GPU, test infrastructure), the same weights
    (int4 x fp16 scale, fp16 embedding / lm_head / cos-sin table taken as the model's parameters), the same op order
    (quarot_llama.py:363-392), and NO fp16 rounding between stages -- what every fp16 implementation approximates.
    Returns (normed hidden [T, H], logits [T, V]) as fp64 numpy."""
    import math
    cfg = model.config
    f64 = torch.float64
    H, I, nq, nkv, d = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    T, eps = inp["T"], cfg.rms_norm_eps

    def unpack(w):       # [N, K/2] int8 -> [N, K] fp64 (low nibble = even k, two's complement)
        u = w.view(torch.uint8).to(torch.int16)
        lo, hi = u & 0xF, u >> 4
        q = torch.stack((lo, hi), dim=-1).reshape(w.shape[0], -1)
        return torch.where(q >= 8, q - 16, q).to(f64)

    def linear(x, lin):  # x @ (w * s)^T, row blocks to bound the fp64 weight copy
        out = torch.empty(x.shape[0], lin.weight.shape[0], dtype=f64, device=DEV)
        sc = lin._scales().to(f64)
        for n0 in range(0, lin.weight.shape[0], 4096):
            w = unpack(lin.weight[n0:n0 + 4096])
            out[:, n0:n0 + 4096] = (x @ w.t()) * sc[n0:n0 + 4096]
        return out

    def ln(h):
        mean = h.mean(-1, keepdim=True)
        var = ((h - mean) ** 2).mean(-1, keepdim=True)
        return (h - mean) / torch.sqrt(var + eps)

    def sylvester(n):
        Hm = torch.ones(1, 1, dtype=f64, device=DEV)
        while Hm.shape[0] < n:
            Hm = torch.cat((torch.cat((Hm, Hm), 1), torch.cat((Hm, -Hm), 1)), 0)
        return Hm
    assert model.head_had_K == 1
    Hh = sylvester(nq) / math.sqrt(nq)
    P = I // model.had_K
    Hp = sylvester(P)
    hadK = model.had_rem_dim.to(f64) if model.had_rem_dim is not None else None
    cs = model.cos_sin_cache.to(f64)
    pos = inp["pos_t"]
    cos, sin = cs[pos, :d // 2], cs[pos, d // 2:]
    hidden = model.embed_tokens[inp["ids_t"]].to(f64)
    bs = inp["kv_np"][0][0].shape[1]
    q_len = T // len(inp["ctx"])
    for li, layer in enumerate(model.layers):
        qkv = linear(ln(hidden), layer.qkv_proj)
        q = qkv[:, :nq * d].reshape(T, nq, d)
        k = qkv[:, nq * d:(nq + nkv) * d].reshape(T, nkv, d)
        v = qkv[:, (nq + nkv) * d:].reshape(T, nkv, d)

        def rope(x):     # neox style: pairs (i, i + d/2)
            x1, x2 = x[..., :d // 2], x[..., d // 2:]
            c, s_ = cos[:, None, :], sin[:, None, :]
            return torch.cat((x1 * c - x2 * s_, x2 * c + x1 * s_), -1)
        q, k = rope(q), rope(k)
        kc = torch.from_numpy(inp["kv_np"][li][0]).to(DEV).to(f64).reshape(-1, nkv, d)     # the fp16 HISTORY is given data
        vc = torch.from_numpy(inp["kv_np"][li][1]).to(DEV).to(f64).reshape(-1, nkv, d)
        slots = torch.from_numpy(inp["slots"]).to(DEV)
        kc[slots], vc[slots] = k, v                                                          # this pass's rows: un-rounded
        attn = torch.empty(T, nq, d, dtype=f64, device=DEV)
        for b, c in enumerate(inp["ctx"].tolist()):
            positions = np.arange(c)
            sl = torch.from_numpy(inp["bt"][b, positions // bs].astype(np.int64) * bs + positions % bs).to(DEV)
            kb = kc[sl].repeat_interleave(nq // nkv, dim=1)        # [c, nq, d]
            vb = vc[sl].repeat_interleave(nq // nkv, dim=1)
            for i in range(q_len):
                t = b * q_len + i
                n = c - q_len + i + 1
                sc_ = torch.einsum("hd,nhd->hn", q[t], kb[:n]) * model.sm_scale
                attn[t] = torch.einsum("hn,nhd->hd", torch.softmax(sc_, -1), vb[:n])
        a = torch.einsum("gh,thd->tgd", Hh, attn).reshape(T, nq * d)
        hidden = hidden + linear(a, layer.o_proj)
        gu = linear(ln(hidden), layer.gate_up)
        up, gate = gu[:, :I], gu[:, I:]
        act = (gate / (1.0 + torch.exp(-gate))) * up
        y = act.reshape(T, model.had_K, P) @ Hp.t()
        if hadK is not None:
            y = torch.einsum("ik,tkj->tij", hadK, y)
        g = y.reshape(T, I) / math.sqrt(I)
        hidden = hidden + linear(g, layer.down_proj)
    normed = ln(hidden)
    logits = normed @ model.lm_head.to(f64).t()
    return normed.cpu().numpy(), logits.cpu().numpy()


def test_full_depth_distance_to_true_math(oracle):
    """North_star's "logits within 1e-3" end to end, stated against the TRUTH instead of against an oracle-vs-oracle floor
    (VERDICT r3): the full-depth verify forward (32 layers, V = 128256, T = 16) in un-rounded fp64 math, and
    |HIP - true| <= 1.1 x |oracle - true| at the median and the 99 % quantile, for the normed hidden state and the logits.
    Both fp16 pipelines sit a few 1e-3 from the truth (32 layers of fp16 roundings on the residual stream); the claim is
    that the HIP path is not further from it than the reference arithmetic's restatement is."""
    from oracle.model import OracleModel
    from qspec_amd.model import CONFIGS, QuarotLlamaForCausalLM, Scratch
    cfg = CONFIGS["llama-3-8b"]
    model = QuarotLlamaForCausalLM(cfg, DEV).init_synthetic(seed=0)
    rng = np.random.default_rng(8)
    ctx_lens, q_len = [40, 130, 9, 260], 4
    inp = make_inputs(model, rng, ctx_lens, q_len)
    true_h, true_l = _true_math_verify_forward(model, inp)
    om = OracleModel.from_torch_model(model, 16)
    kv_np = [(k.copy(), v.copy()) for k, v in inp["kv_np"]]
    ref = om.forward(inp["ids"], inp["pos"], kv_np, inp["slots"], inp["bt"], inp["ctx"], inp["q_start"], False)
    ref_l = om.logits(ref)
    s = Scratch(cfg, inp["T"], len(ctx_lens), q_len, inp["n_splits"], DEV)
    out = model.forward(inp["ids_t"], inp["pos_t"], inp["kv_t"], inp["md"], s, w4a4=False)
    logits = model.compute_logits(out, s)
    torch.cuda.synchronize()
    for what, hip, orc, tru in (("normed hidden", out.cpu().numpy(), ref, true_h), ("logits", logits.cpu().numpy(), ref_l, true_l)):
        e_hip, e_or = _rel3(hip, tru), _rel3(orc, tru)
        qh, qo = np.quantile(e_hip, [0.5, 0.99]), np.quantile(e_or, [0.5, 0.99])
        print(f"full depth, verify T=16, {what}: |HIP - true| / 1e-3 median {qh[0]:.2f} q99 {qh[1]:.2f} max {e_hip.max():.1f};  "
              f"|oracle - true| median {qo[0]:.2f} q99 {qo[1]:.2f} max {e_or.max():.1f};  |HIP - oracle| median "
              f"{np.median(_rel3(hip, orc)):.2f}")
        assert qh[0] <= 1.1 * qo[0] and qh[1] <= 1.1 * qo[1], (what, qh, qo)
    # the greedy token of both pipelines agrees with the truth's wherever the truth's two best logits are not a near tie
    top2 = np.sort(true_l, -1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 0.05
    assert np.array_equal(logits.float().argmax(-1).cpu().numpy()[clear], true_l.argmax(-1)[clear])
    assert np.array_equal(ref_l.astype(np.float32).argmax(-1)[clear], true_l.argmax(-1)[clear])


def test_draft_divergence_is_the_attention_outputs_last_bit(oracle):
    """The cause behind the statistical (not exact) bars on end-to-end draft distributions, shown directly: a
    3-layer model at the Llama-3-8B width, 32 decode tokens over a random paged KV history (the config-3 draft shape), three
    input draws.  The attention output is the ONE stage of the draft layer whose bits the hardware fixes (2^x instruction, MFMA
    summation order): un-forced, a few hundred of its 131 k fp16 values per layer differ from the oracle's in the last bit
    (asserted: within 1e-3, and not all equal); most of those differences are absorbed by the head Hadamard + int4 quantiser,
    the rest is what occasionally flips an int4 abs-max and a whole row with it.  With NOTHING but each layer's attention
    output teacher-forced from the oracle, every row of the final hidden state and every K / V row written are bit-identical
    to the oracle (asserted for every draw): every other stage is exact, so attention's last bit is the whole difference."""
    from oracle.model import OracleModel
    from qspec_amd import ops as _ops
    from qspec_amd.model import QuarotLlamaConfig, QuarotLlamaForCausalLM
    cfg = QuarotLlamaConfig(4096, 14336, 32, 8, 3, 2048, 1e-5, 500000.0, 1024, "llama-3-8b-3layer")
    model = QuarotLlamaForCausalLM(cfg, DEV).init_synthetic(seed=3, lm_head_std=0.05)
    om = OracleModel.from_torch_model(model, 16)
    total_rows = total_bits = 0
    for draw in range(3):
        rng = np.random.default_rng(31 + draw)
        ctx_lens = [int(c) for c in rng.integers(40, 500, 32)]
        inp = make_inputs(model, rng, ctx_lens, 1)
        kv_np = [(k.copy(), v.copy()) for k, v in inp["kv_np"]]
        ref, trace = om.forward(inp["ids"], inp["pos"], kv_np, inp["slots"], inp["bt"], inp["ctx"], inp["q_start"], True,
                                return_trace=True)
        grabbed = []
        real = _ops.paged_attention

        def spy(q, q_stride, kc, vc, bt, ctx, qs, tokens, max_q, nh, sm, ns, ws, out):
            real(q, q_stride, kc, vc, bt, ctx, qs, tokens, max_q, nh, sm, ns, ws, out)
            grabbed.append(out.clone())

        def run(override, watch=False):
            kv = [(torch.from_numpy(k).to(DEV), torch.from_numpy(v).to(DEV)) for k, v in inp["kv_np"]]
            if watch:
                _ops.paged_attention = spy
            try:
                out = model.forward_modulewise(inp["ids_t"], inp["pos_t"], kv, inp["md"], w4a4=True, attn_override=override)
                torch.cuda.synchronize()
            finally:
                _ops.paged_attention = real
            return out.cpu().numpy(), kv
        plain, _ = run(None, watch=True)
        forced, kv_f = run([torch.from_numpy(trace[f"attn_{li}"]).to(DEV) for li in range(cfg.num_hidden_layers)])
        assert np.array_equal(forced.view(np.uint16), ref.view(np.uint16)), draw        # every row bit-identical
        for li in range(cfg.num_hidden_layers):
            assert np.array_equal(kv_f[li][0].cpu().numpy().view(np.uint16), kv_np[li][0].view(np.uint16)), (draw, li)
            assert np.array_equal(kv_f[li][1].cpu().numpy().view(np.uint16), kv_np[li][1].view(np.uint16)), (draw, li)
        # layer 0's attention has identical inputs on both sides: within 1e-3 of the oracle's
        a_hip, a_ref = grabbed[0].cpu().numpy(), trace["attn_0"]
        assert np.abs(a_hip.astype(np.float64) - a_ref.astype(np.float64)).max() <= 1e-3 * max(1.0, float(np.abs(a_ref).max()))
        total_bits += int((a_hip.view(np.uint16) != a_ref.view(np.uint16)).sum())
        total_rows += int((plain.view(np.uint16) != ref.view(np.uint16)).any(axis=1).sum())
    print(f"3 draws x 32 rows x 3 layers: layer-0 attention outputs differing in the last bit(s): {total_bits} of {3 * 32 * 4096}; "
          f"rows of the final hidden state differing un-forced: {total_rows} of 96; with attention forced: 0")
    assert total_bits > 0        # the difference this test is about exists


def test_engine_cycle_with_typical_acceptance_sampler(tiny, oracle):
    """The cycle with draft_token_acceptance_method = typical_acceptance_sampler: the engine's output must be what the
    reference rule (the oracle's restatement, pinned to the reference class by tests/golden/typical_acceptance.npz) gives on
    the GPU's own target distribution and draft tokens; the captured graph replays it; counters follow."""
    from qspec_amd.spec_decode import QSpecEngine, TypicalAcceptanceSampler
    rng = np.random.default_rng(21)
    prompts = [rng.integers(0, tiny.config.vocab_size, n).tolist() for n in (20, 31, 8, 50)]
    gens = []
    for use_graph in (False, True):
        eng = QSpecEngine(tiny, 3, 4, max_model_len=256, block_size=16, max_new_tokens=64, use_graph=use_graph, seed=5,
                          acceptance_sampler=TypicalAcceptanceSampler(0.09, 0.3))
        eng.add_sequences(prompts)
        for _ in range(4):
            eng.step()
            torch.cuda.synchronize()
            out, acc, rec, c, _ = oracle.typical_acceptance_sample(eng.target_probs.cpu().numpy(), eng.target_tokens[:, 3].cpu().numpy(),
                                                                   eng.draft_ids_kb.t().cpu().numpy(), 0.09, 0.3)
            assert np.array_equal(eng.out_tokens.cpu().numpy(), out)
            assert np.array_equal(eng.accepted.cpu().numpy().astype(bool), acc)
        m = eng.metrics()
        assert m.draft_tokens == 4 * 4 * 3 and m.emitted_tokens >= 16
        gens.append(eng.generated())
    assert gens[0] == gens[1]


def test_engine_cycle_with_sampled_requests(tiny, oracle):
    """A mixed batch: slots 0 / 1 greedy, slots 2 / 3 sampling (temperature, top-k, top-p).  The whole cycle then goes through
    the sampling front end (ops.sample_top_k_top_p): every distribution the rejection sampler sees is the PROCESSED one (sums to
    1, at most top_k non-zeros for the sampled rows), draft tokens lie in its support, greedy rows propose their argmax, the
    output follows _create_output on the GPU's own accept mask, the captured graph replays the eager cycle token for token,
    and a batch that is greedy again goes back to the greedy graph."""
    from qspec_amd.spec_decode import QSpecEngine
    rng = np.random.default_rng(23)
    prompts = [rng.integers(0, tiny.config.vocab_size, n).tolist() for n in (20, 31, 8, 50)]
    gens = []
    for use_graph in (False, True):
        eng = QSpecEngine(tiny, 3, 4, max_model_len=256, block_size=16, max_new_tokens=64, use_graph=use_graph, seed=9)
        eng.set_sampling_params(2, temperature=0.8, top_k=50, top_p=0.9)
        eng.set_sampling_params(3, temperature=1.2, top_k=-1, top_p=0.5)
        eng.add_sequences(prompts)
        for _ in range(4):
            eng.step()
            torch.cuda.synchronize()
            assert eng._mode_sampling
            dp = eng.draft_probs_kbv.transpose(0, 1).cpu().numpy()        # [B, k, V]
            tp = eng.target_probs.cpu().numpy()
            ids = eng.draft_ids_kb.t().cpu().numpy()
            assert np.allclose(dp.sum(-1), 1.0, atol=1e-4) and np.allclose(tp.sum(-1), 1.0, atol=1e-4)
            assert ((dp[2] > 0).sum(-1) <= 50).all() and ((tp[2] > 0).sum(-1) <= 50).all()
            assert ((dp[3] > 0).sum(-1) < tiny.config.vocab_size).all()      # top-p 0.5 cuts something
            for b in range(4):
                for i in range(3):
                    assert dp[b, i, ids[b, i]] > 0
            assert np.array_equal(ids[:2], dp[:2].argmax(-1))                 # greedy rows of the mixed batch
            out, _ = oracle.create_output(eng.accepted.cpu().numpy().astype(bool), eng.recovered.cpu().numpy(), ids,
                                          eng.target_tokens[:, 3].cpu().numpy())
            assert np.array_equal(eng.out_tokens.cpu().numpy(), out)
        gens.append(eng.generated())
        if use_graph:
            assert eng._graph_s is not None and eng._graph is None
            eng.set_sampling_params(2); eng.set_sampling_params(3)            # all greedy again: the greedy cycle and its graph
            eng.step()
            torch.cuda.synchronize()
            assert not eng._mode_sampling and eng._graph is not None
    assert gens[0] == gens[1]


def test_graph_replay_equals_eager(tiny):
    """The captured hipGraph of the cycle produces the same tokens as the eager cycle (same Philox stream)."""
    from qspec_amd.spec_decode import QSpecEngine
    rng = np.random.default_rng(3)
    prompts = [rng.integers(0, tiny.config.vocab_size, n).tolist() for n in (20, 31, 8, 50)]
    outs = []
    for use_graph in (False, True):
        eng = QSpecEngine(tiny, 3, 4, max_model_len=256, block_size=16, max_new_tokens=64, use_graph=use_graph, seed=5)
        eng.add_sequences(prompts)
        for _ in range(6):
            eng.step()
        outs.append((eng.generated(), eng.metrics()))
    assert outs[0][0] == outs[1][0]
    assert outs[0][1] == outs[1][1]
    assert all(len(g) >= 7 for g in outs[0][0])  # at least one token per cycle + the prefill token


def test_worker_api(tiny):
    """create_spec_worker / execute_model contract (spec_decode_worker.py:53-113,461-560,972-1063)."""
    from qspec_amd.spec_decode import ExecuteModelRequest, SequenceGroupMetadata, create_spec_worker
    from qspec_amd.spec_decode.worker import SequenceData, SpeculativeConfig
    rng = np.random.default_rng(4)
    w = create_spec_worker(model_config=tiny.config, model=tiny, speculative_config=SpeculativeConfig(3),
                           max_num_seqs=4, max_model_len=256, block_size=16, device=DEV)
    w.init_device()
    assert w.proposer_model is w.scorer_model                     # shared module
    nb, _ = w.determine_num_available_blocks()                    # profiled: a prompt pass + one captured cycle
    prof = w.memory_profile
    assert nb > 64 and nb * prof["cache_block_size"] <= prof["free"] and prof["torch_peak_increase"] > 0
    assert prof["available_kv_cache_memory"] <= prof["total"] * 0.9 - (prof["total"] - prof["free"]) - 3 * prof["torch_peak_increase"] + 1
    w.initialize_cache(64, 0)                                     # (not the whole card: other tests share it)
    assert w.engine.num_blocks == 64 and w.engine.kv_caches[0][0].shape[0] == 64
    with pytest.raises(NotImplementedError):
        w.get_cache_block_size_bytes()
    sg = [SequenceGroupMetadata(f"r{i}", True, {i: SequenceData(rng.integers(0, 2048, 10 + i).tolist())}) for i in range(4)]
    out = w.execute_model(ExecuteModelRequest(sg, num_lookahead_slots=0))
    assert len(out) == 1 and len(out[0].outputs) == 4 and all(t >= 0 for t in out[0].token_ids())
    assert [o.samples[0].parent_seq_id for o in out[0].outputs] == [0, 1, 2, 3]
    for s in sg:
        s.is_prompt = False
    req = ExecuteModelRequest(sg, num_lookahead_slots=3)
    outs = w.execute_model(req)
    assert req.w4a4 is False and w.proposer_calls == 3 and w.scorer_calls == 2
    assert 1 <= len(outs) <= 4
    toks = torch.tensor([o.token_ids() for o in outs]).t()        # [B, steps]
    assert (toks[:, 0] != -1).all()                               # every sequence emits at least one token
    for row in toks.tolist():                                     # -1 only as a suffix
        seen = False
        for t in row:
            assert not (seen and t != -1)
            seen = seen or t == -1
    assert w.execute_model(None) == []


# ------------------------------------------------------------------ the other model families of BASELINE.json's configs

FAMILIES = {
    # full layer width of the named config, one layer, small vocabulary (the layer shapes are what differs)
    "llama-3-8b": (4096, 14336, 32, 8, 500000.0),        # the headline model of bench.py (configs 2 and 3)
    "tinyllama-1.1b": (2048, 5632, 32, 4, 10000.0),      # head_dim 64 (generic attention), I = had44 (x) H128
    "llama-2-13b": (5120, 13824, 40, 40, 10000.0),       # 40 heads = had40 on the head axis, I = had108 (x) H128
    "llama-3-70b": (8192, 28672, 64, 8, 500000.0),       # 64 heads (FWHT-64), I = had28 (x) H1024
}


@pytest.mark.parametrize("family", list(FAMILIES))
@pytest.mark.parametrize("w4a4", [True, False])
def test_model_family_layer_matches_oracle(oracle, family, w4a4):
    from oracle.model import OracleModel
    from qspec_amd.model import QuarotLlamaConfig, QuarotLlamaForCausalLM, Scratch
    H, I, nh, nkv, theta = FAMILIES[family]
    cfg = QuarotLlamaConfig(H, I, nh, nkv, 1, 1024, 1e-5, theta, 512, family + "-1layer")
    model = QuarotLlamaForCausalLM(cfg, DEV).init_synthetic(seed=2, lm_head_std=0.05)
    rng = np.random.default_rng(4)
    ctx_lens, q_len = ([33, 150], 1) if w4a4 else ([33, 150], 2)
    inp = make_inputs(model, rng, ctx_lens, q_len)
    om = OracleModel.from_torch_model(model, 16)
    kv_np = [(k.copy(), v.copy()) for k, v in inp["kv_np"]]
    ref = om.forward(inp["ids"], inp["pos"], kv_np, inp["slots"], inp["bt"], inp["ctx"], inp["q_start"], w4a4)
    s = Scratch(cfg, inp["T"], len(ctx_lens), q_len, inp["n_splits"], DEV)
    out = model.forward(inp["ids_t"], inp["pos_t"], inp["kv_t"], inp["md"], s, w4a4=w4a4)
    torch.cuda.synchronize()
    diff = np.abs(out.cpu().numpy().astype(np.float64) - ref.astype(np.float64))
    if w4a4:
        assert np.median(diff) < 1e-3 and np.quantile(diff, 0.99) < 0.1, (np.median(diff), np.quantile(diff, 0.99))
        k_hip = inp["kv_t"][0][0].cpu().numpy()
        assert np.array_equal(k_hip.view(np.uint16), kv_np[0][0].view(np.uint16))
    else:
        floor, rel = _w4a16_noise_floor(om, inp, ref), _rel3(out.cpu().numpy(), ref)
        print(f"{family} w4a16 layer, err / 1e-3: max {rel.max():.2f} q99 {np.quantile(rel, 0.99):.2f}; floor max {floor.max():.2f} q99 {np.quantile(floor, 0.99):.2f}")
        assert np.quantile(rel, 0.99) < max(1.5 * np.quantile(floor, 0.99), 2.0) and rel.max() < max(2.0 * floor.max(), 4.0), \
            (np.quantile(rel, 0.99), np.quantile(floor, 0.99), rel.max(), floor.max())
    # and the reference-order module-wise path gives the same bits as the fused one
    kv_b = [(torch.from_numpy(k).to(DEV), torch.from_numpy(v).to(DEV)) for k, v in inp["kv_np"]]
    b = model.forward_modulewise(inp["ids_t"], inp["pos_t"], kv_b, inp["md"], w4a4=w4a4)
    torch.cuda.synchronize()
    if w4a4:
        assert torch.equal(out.view(torch.int16), b.view(torch.int16))
    else:
        assert (out.float() - b.float()).abs().max().item() < 2e-2


@pytest.mark.parametrize("family,ctx_lens,q_len", [("llama-3-8b", [40, 130, 9, 260], 4), ("llama-3-8b", [77], 4),
                                                   ("llama-2-13b", [33, 150, 20], 4), ("llama-3-70b", [33, 150], 4)])
def test_fragment_major_activation_tiles_change_no_bit(family, ctx_lens, q_len, monkeypatch):
    """Verify pass at <= 16 tokens: norm / head transform / MLP transform storing fragment-major tiles and the W4A16 GEMMs
    loading their operands straight from them (model.ACT_FRAGMENT_MAJOR) against the row-major buffers + LDS regrouping:
    the normed hidden state and every KV slot bit for bit, two layers deep (the second norm takes the K-slice finish).
    Llama-3-70B's K = 8192 has no tile form: the switch must then be inert."""
    from qspec_amd.model import QuarotLlamaConfig, QuarotLlamaForCausalLM, Scratch
    H, I, nh, nkv, theta = FAMILIES[family]
    cfg = QuarotLlamaConfig(H, I, nh, nkv, 2, 1024, 1e-5, theta, 512, family + "-2layer")
    model = QuarotLlamaForCausalLM(cfg, DEV).init_synthetic(seed=5, lm_head_std=0.05)
    rng = np.random.default_rng(6)
    inp = make_inputs(model, rng, ctx_lens, q_len)
    s = Scratch(cfg, inp["T"], len(ctx_lens), q_len, inp["n_splits"], DEV)
    assert model._fragment_major_ok(inp["T"], inp["md"], s) == (family != "llama-3-70b")
    outs = []
    for on in (True, False):
        monkeypatch.setattr(model, "ACT_FRAGMENT_MAJOR", on)
        kv = [(k.clone(), v.clone()) for k, v in inp["kv_t"]]
        outs.append((model.forward(inp["ids_t"], inp["pos_t"], kv, inp["md"], s, w4a4=False).clone(), kv))
    torch.cuda.synchronize()
    (a, kva), (b, kvb) = outs
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    for (ka, va), (kb, vb) in zip(kva, kvb):
        assert torch.equal(ka, kb) and torch.equal(va, vb)


@pytest.mark.parametrize("family,n_seqs,q_len", [("llama-3-8b", 8, 4), ("llama-3-8b", 5, 4), ("llama-3-70b", 8, 4), ("llama-2-13b", 6, 4)])
def test_two_tile_verify_layers_at_17_to_32_tokens(oracle, family, n_seqs, q_len, monkeypatch):
    """Verify pass at 17..32 tokens: qkv / o_proj / gate_up on the two-token-tile W4A16 streaming kernel over two fragment-major
    tiles (model.forward picks it where every layer shape has the form) against the M-tiled path (ACT_FRAGMENT_MAJOR off): two
    fp32 summation orders, so the normed hidden state within the W4A16 bar of the fused-vs-module-wise tests and the KV rows
    within 1e-3; and against the oracle within the noise floor, as test_model_family_layer_matches_oracle does."""
    from oracle.model import OracleModel
    from qspec_amd.model import QuarotLlamaConfig, QuarotLlamaForCausalLM, Scratch
    H, I, nh, nkv, theta = FAMILIES[family]
    cfg = QuarotLlamaConfig(H, I, nh, nkv, 1, 1024, 1e-5, theta, 512, family + "-1layer")
    model = QuarotLlamaForCausalLM(cfg, DEV).init_synthetic(seed=9, lm_head_std=0.05)
    rng = np.random.default_rng(10)
    ctx_lens = [33 + 17 * i for i in range(n_seqs)]
    inp = make_inputs(model, rng, ctx_lens, q_len)
    T = inp["T"]
    assert 16 < T <= 32
    s = Scratch(cfg, T, n_seqs, q_len, inp["n_splits"], DEV)
    outs = []
    for on in (True, False):
        monkeypatch.setattr(model, "ACT_FRAGMENT_MAJOR", on)
        kv = [(k.clone(), v.clone()) for k, v in inp["kv_t"]]
        outs.append((model.forward(inp["ids_t"], inp["pos_t"], kv, inp["md"], s, w4a4=False).clone(), kv))
        assert model.last_forward_form == ("two-tile" if on else "unfused")
    torch.cuda.synchronize()
    (a, kva), (b, kvb) = outs
    assert (a.float() - b.float()).abs().max().item() < 2e-2
    for (ka, va), (kb, vb) in zip(kva, kvb):
        assert (ka.float() - kb.float()).abs().max().item() <= 1e-3 * max(1.0, kb.float().abs().max().item())
        assert (va.float() - vb.float()).abs().max().item() <= 1e-3 * max(1.0, vb.float().abs().max().item())
    om = OracleModel.from_torch_model(model, 16)
    kv_np = [(k.copy(), v.copy()) for k, v in inp["kv_np"]]
    ref = om.forward(inp["ids"], inp["pos"], kv_np, inp["slots"], inp["bt"], inp["ctx"], inp["q_start"], False)
    floor, rel = _w4a16_noise_floor(om, inp, ref), _rel3(a.cpu().numpy(), ref)
    assert np.quantile(rel, 0.99) < max(1.5 * np.quantile(floor, 0.99), 2.0) and rel.max() < max(2.0 * floor.max(), 4.0), \
        (np.quantile(rel, 0.99), np.quantile(floor, 0.99), rel.max(), floor.max())


def test_verify_o_proj_k_sliced_dev_knob_meets_the_same_bar(oracle, monkeypatch):
    """QSPEC_VERIFY_O_SLICES (model.py, dev knob; measured in rounds 1 and 3 and left off: DESIGN.md section 4): the verify pass's
    o_proj as K slices whose raw fp32 sums the following norm finishes, as down_proj's are.  Same comparison and bars as
    test_model_family_layer_matches_oracle at the Llama-3-8B width, and within two fp16 ulps of the unsliced form."""
    from oracle.model import OracleModel
    from qspec_amd.model import QuarotLlamaConfig, QuarotLlamaForCausalLM, Scratch
    H, I, nh, nkv, theta = FAMILIES["llama-3-8b"]
    cfg = QuarotLlamaConfig(H, I, nh, nkv, 1, 1024, 1e-5, theta, 512, "llama-3-8b-1layer")
    model = QuarotLlamaForCausalLM(cfg, DEV).init_synthetic(seed=2, lm_head_std=0.05)
    rng = np.random.default_rng(4)
    inp = make_inputs(model, rng, [33, 150], 2)
    om = OracleModel.from_torch_model(model, 16)
    ref = om.forward(inp["ids"], inp["pos"], [(k.copy(), v.copy()) for k, v in inp["kv_np"]], inp["slots"], inp["bt"], inp["ctx"],
                     inp["q_start"], False)
    outs = {}
    for S in (0, 2, 4):
        monkeypatch.setattr(QuarotLlamaForCausalLM, "VERIFY_O_SLICES", S)
        kv = [(torch.from_numpy(k).to(DEV), torch.from_numpy(v).to(DEV)) for k, v in inp["kv_np"]]
        s = Scratch(cfg, inp["T"], 2, 2, inp["n_splits"], DEV)
        outs[S] = model.forward(inp["ids_t"], inp["pos_t"], kv, inp["md"], s, w4a4=False).float().cpu().numpy()
    floor = _w4a16_noise_floor(om, inp, ref)
    for S in (2, 4):
        rel = _rel3(outs[S], ref)
        assert np.quantile(rel, 0.99) < max(1.5 * np.quantile(floor, 0.99), 2.0) and rel.max() < max(2.0 * floor.max(), 4.0)
        assert _rel3(outs[S], outs[0]).max() < 4.0


def test_worker_decode_step_with_speculation_disabled(tiny):
    """num_lookahead_slots == 0 on a decode batch (or speculative_disable_by_batch_size reached): the scorer alone runs,
    W4A16, one token per sequence (spec_decode_worker.py:497-538,666-720).  The token is cross-checked against the
    module-wise (reference op order) forward on a copy of the KV cache: same positions, slots and context lengths."""
    from qspec_amd.model import AttentionMetadata
    from qspec_amd.spec_decode import ExecuteModelRequest, SequenceGroupMetadata, create_spec_worker
    from qspec_amd.spec_decode.worker import SequenceData, SpeculativeConfig
    rng = np.random.default_rng(9)
    w = create_spec_worker(model_config=tiny.config, model=tiny, speculative_config=SpeculativeConfig(3, speculative_disable_by_batch_size=2),
                           max_num_seqs=4, max_model_len=256, block_size=16, device=DEV)
    w.init_device()
    w.initialize_cache(64, 0)
    lens = [10, 23, 5, 40]
    sg = [SequenceGroupMetadata(f"r{i}", True, {i: SequenceData(rng.integers(0, 2048, n).tolist())}) for i, n in enumerate(lens)]
    w.execute_model(ExecuteModelRequest(sg, num_lookahead_slots=0))
    for s in sg:
        s.is_prompt = False
    eng = w.engine
    seq0, last0 = eng.seq_lens.clone(), eng.last_token.clone()
    # expectation from the module-wise path on a copy of the cache
    kv = [(k.clone(), v.clone()) for k, v in eng.kv_caches]
    pos = (seq0 - 1).to(torch.int64)
    slots = torch.stack([eng._slots_for(b, pos[b:b + 1])[0] for b in range(4)])
    md = AttentionMetadata(slots, eng.block_tables, seq0.clone(), torch.arange(5, dtype=torch.int32, device=DEV), 1, 1)
    hs = tiny.forward_modulewise(last0, pos, kv, md, w4a4=False)
    from qspec_amd.model import Scratch
    expect = tiny.compute_logits(hs, Scratch(tiny.config, 4, 4, 1, 1, DEV)).float().argmax(-1)
    calls = (w.proposer_calls, w.scorer_calls)
    # (a) the scheduler asks for no lookahead slots; (b) the running queue reaches speculative_disable_by_batch_size
    outs = w.execute_model(ExecuteModelRequest(sg, num_lookahead_slots=0))
    assert len(outs) == 1 and outs[0].token_ids() == expect.cpu().tolist()
    assert (w.proposer_calls, w.scorer_calls) == (calls[0], calls[1] + 1)
    assert torch.equal(eng.seq_lens, seq0 + 1) and torch.equal(eng.last_token, expect)
    outs = w.execute_model(ExecuteModelRequest(sg, num_lookahead_slots=3, running_queue_size=2))
    assert len(outs) == 1 and all(t >= 0 for t in outs[0].token_ids()) and w.proposer_calls == calls[0]
    # and a speculative step still works afterwards
    outs = w.execute_model(ExecuteModelRequest(sg, num_lookahead_slots=3))
    assert 1 <= len(outs) <= 4 and w.proposer_calls == calls[0] + 3


# ------------------------------------------------------------------ teacher-forced layers at the headline model's width

def _ulps16(got, ref):
    """|got - ref| in units of the fp16 spacing at |ref| (subnormal spacing below 2^-14)."""
    got, ref = got.astype(np.float64), ref.astype(np.float64)
    e = np.floor(np.log2(np.maximum(np.abs(ref), 2.0 ** -14)))
    return np.abs(got - ref) / 2.0 ** (e - 10)


@pytest.mark.parametrize("w4a4,q_len,B", [(True, 1, 4), (False, 4, 4), (True, 1, 32), (False, 6, 32)])
def test_teacher_forced_layers_llama3_8b(oracle, w4a4, q_len, B):
    """Per-layer parity without accumulated drift (quarot_llama.py:363-392): a 3-layer oracle model at the full
    Llama-3-8B width runs once and records every layer's input; each HIP layer is then fed THE ORACLE'S input of that
    layer (as the embedding table of a one-layer model sharing the layer's weights and KV cache) and compared with
    the oracle's output of that layer alone.
      verify (W4A16): every element of the layer output within 1e-3 (absolute below 1, relative above) -- north_star's bar;
      draft  (W4A4):  everything up to the attention boundary (norm + int4 quant + qkv GEMM + RoPE + KV write) bit for
                      bit; behind it the int4 pipeline may turn a last-bit attention difference into one different int4
                      somewhere: the o_proj / down_proj input bytes must agree except for a sliver, and the layer output
                      within one int4 step's worth of the residual."""
    from oracle.model import OracleModel
    from qspec_amd.model import AttentionMetadata, QuarotLlamaConfig, QuarotLlamaForCausalLM, Scratch
    L = 3
    cfg = QuarotLlamaConfig(4096, 14336, 32, 8, L, 1024, 1e-5, 500000.0, 1024, "llama-3-8b-3layer")
    model = QuarotLlamaForCausalLM(cfg, DEV).init_synthetic(seed=5, lm_head_std=0.05)
    rng = np.random.default_rng(31 + B + q_len)
    ctx_lens = [int(c) for c in rng.integers(q_len + 1, 400, B)]
    ctx_lens[0], ctx_lens[-1] = q_len + 1, 515
    inp = make_inputs(model, rng, ctx_lens, q_len)
    om = OracleModel.from_torch_model(model, 16)
    kv_np = [(k.copy(), v.copy()) for k, v in inp["kv_np"]]
    _, tr = om.forward(inp["ids"], inp["pos"], kv_np, inp["slots"], inp["bt"], inp["ctx"], inp["q_start"], w4a4,
                       return_trace=True)
    T = inp["T"]
    n_splits = inp["n_splits"]
    cfg1 = QuarotLlamaConfig(4096, 14336, 32, 8, 1, 1024, 1e-5, 500000.0, 1024, "one-layer")
    one = QuarotLlamaForCausalLM(cfg1, DEV)
    ids = torch.arange(T, dtype=torch.int64, device=DEV)
    s = Scratch(cfg1, T, B, q_len, n_splits, DEV)
    failures = []
    for li in range(L):
        h_in = om.embed_tokens[inp["ids"]] if li == 0 else tr[f"hidden_{li - 1}"]
        one.layers = [model.layers[li]]
        one.embed_tokens = torch.from_numpy(np.ascontiguousarray(h_in)).to(DEV)
        kv = [inp["kv_t"][li]]                                       # the random history; the layer writes its q_len slots
        out = one.forward(ids, inp["pos_t"], kv, inp["md"], s, w4a4=w4a4)
        torch.cuda.synchronize()
        hid = s.hidden[:T].cpu().numpy()                            # residual stream after the layer
        ref_hid = tr[f"hidden_{li}"]
        qkv = s.act_buffer_qkv[:T].cpu().numpy()
        k_hip, v_hip = kv[0][0].cpu().numpy(), kv[0][1].cpu().numpy()
        if w4a4:
            assert np.array_equal(qkv.view(np.uint16), tr[f"qkv_{li}"].view(np.uint16)), li
            assert np.array_equal(k_hip.view(np.uint16), kv_np[li][0].view(np.uint16)), li
            assert np.array_equal(v_hip.view(np.uint16), kv_np[li][1].view(np.uint16)), li
            q_o = s.quantized_buffer_qkv[:T].cpu().numpy()
            if T <= 4 and one.HADAMARD_QUANT_IN_OPROJ:
                # the o_proj launch quantises the fp16 head-Hadamard rows in its own prologue (bit-identical to the
                # Quantizer: test_spread_head_hadamard_and_quantiser_in_the_o_proj_prologue): its input bytes are those
                q_o = np.asarray(oracle.rowabsmax_quant_i4(s.act_buffer_had[:T].cpu().numpy(), 1.0)[0])
            q_d = s.quantized_buffer_mlp[:T].cpu().numpy()
            # (batch 32 takes the separate-norm branch, where the post-attention norm reuses this buffer)
            f_o = float((q_o != tr[f"o_in_{li}"][0]).mean()) if B <= 16 else 0.0
            f_d = float((q_d != tr[f"down_in_{li}"][0]).mean())
            u = _ulps16(hid, ref_hid)
            print(f"layer {li} w4a4 B={B}: differing int4 bytes o_proj in {f_o:.2e}, down_proj in {f_d:.2e}; "
                  f"hidden ulps median {np.median(u):.2f} q99 {np.quantile(u, 0.99):.1f} max {u.max():.1f}")
            assert f_o < 5e-3 and f_d < 2e-2, (li, f_o, f_d)
            err = np.abs(hid.astype(np.float64) - ref_hid.astype(np.float64))
            bar = 1e-3 * np.maximum(1.0, np.abs(ref_hid.astype(np.float64)))
            assert np.mean(err > bar) < 2e-2 and np.quantile(err / bar, 0.999) < 8.0, (li, np.mean(err > bar), (err / bar).max())
        else:
            def close(got, ref, what, max_bar):
                got, ref = got.astype(np.float64), ref.astype(np.float64)
                r = np.abs(got - ref) / (1e-3 * np.maximum(1.0, np.abs(ref)))
                print(f"layer {li} w4a16 B={B} {what}: max err/1e-3 {r.max():.2f}, above 1e-3: {(r > 1).mean():.2e}")
                if r.max() > max_bar or (r > 1).mean() > 1e-3:
                    failures.append((li, what, float(r.max()), float((r > 1).mean())))
            # The qkv GEMM consumes identical inputs: within 1e-3 of the fp64-accumulate oracle before RoPE (<= 1 fp16
            # ulp), 2 ulps for a handful of elements behind RoPE (two rounded operands per output).  Behind that the
            # layer is a chain of ~10 fp16 roundings whose 1-ulp flips are mixed by the Hadamards / the residual add
            # (an outlier's ulp lands on every element of its row), so the layer output is compared after the norm:
            # >= 90 % of the elements within 1e-3, all within 1e-2.  Stage by stage on identical inputs every launch
            # meets 1e-3: test_verify_layer_stagewise_teacher_forced below.
            close(qkv, tr[f"qkv_{li}"], "qkv", 2.5)
            close(k_hip, kv_np[li][0], "key cache", 2.5)
            got, ref = out.cpu().numpy().astype(np.float64), oracle.ln_fp16(ref_hid, 1e-5).astype(np.float64)
            r = np.abs(got - ref) / (1e-3 * np.maximum(1.0, np.abs(ref)))
            print(f"layer {li} w4a16 B={B} normed layer output: max err/1e-3 {r.max():.2f}, above 1e-3: {(r > 1).mean():.2e}")
            if r.max() > 10.0 or (r > 1).mean() > 0.1:
                failures.append((li, "normed output", float(r.max()), float((r > 1).mean())))
    assert not failures, failures


@pytest.mark.parametrize("B,q_len", [(4, 4), (32, 6)])
def test_verify_layer_stagewise_teacher_forced(oracle, B, q_len):
    """Every launch of a VERIFY layer at the Llama-3-8B width (T = 16: streaming kernels and fused epilogues; T = 192:
    M-tiled kernels, config 3) fed with the ORACLE's input of that stage and compared with the oracle's output of
    that stage -- so each 1e-3 claim is made on identical inputs, with nothing accumulated
    (quarot_llama.py:363-392; the W4A16 oracle accumulates in fp64)."""
    from oracle.model import OracleModel
    from qspec_amd import ops
    from qspec_amd.model import QuarotLlamaConfig, QuarotLlamaForCausalLM
    cfg = QuarotLlamaConfig(4096, 14336, 32, 8, 2, 1024, 1e-5, 500000.0, 1024, "llama-3-8b-2layer")
    model = QuarotLlamaForCausalLM(cfg, DEV).init_synthetic(seed=6, lm_head_std=0.05)
    rng = np.random.default_rng(77 + B)
    ctx_lens = [int(c) for c in rng.integers(q_len + 1, 400, B)]
    ctx_lens[0], ctx_lens[-1] = q_len + 1, 515
    inp = make_inputs(model, rng, ctx_lens, q_len)
    om = OracleModel.from_torch_model(model, 16)
    kv_before = [(k.copy(), v.copy()) for k, v in inp["kv_np"]]
    kv_np = [(k.copy(), v.copy()) for k, v in inp["kv_np"]]
    _, tr = om.forward(inp["ids"], inp["pos"], kv_np, inp["slots"], inp["bt"], inp["ctx"], inp["q_start"], False,
                       return_trace=True)
    T, md = inp["T"], inp["md"]
    H, I, nq, nkv, d = 4096, 14336, 32, 8, 128
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    e16 = lambda *s: torch.empty(*s, dtype=torch.float16, device=DEV)  # noqa: E731
    report = []

    def check(what, got, ref, bar=1.0, scale_rows=False):
        got, ref = got.astype(np.float64), np.asarray(ref).astype(np.float64)
        den = np.maximum(1.0, np.abs(ref))
        if scale_rows:   # outputs that are sums with cancellation: relative to the row's largest operand
            den = np.maximum(den, np.abs(ref).max(axis=-1, keepdims=True) * 0 + scale_rows)
        r = np.abs(got - ref) / (1e-3 * den)
        report.append((what, float(r.max()), float((r > 1).mean())))
        print(f"B={B} {what}: max err/1e-3 {r.max():.3f}, above 1e-3: {(r > 1).mean():.2e}")
        assert r.max() <= bar, (what, r.max())

    li = 1                                                      # layer 1: its input has been through a full layer
    Lw = model.layers[li]
    hid_in = tr[f"hidden_{li - 1}"]
    # 1. input norm (fp16 mode): exact
    normed = e16(T, H)
    ops.add_rms_norm_fp16(normed, None, t(hid_in), None, 1e-5)
    torch.cuda.synchronize()
    assert np.array_equal(normed.cpu().numpy().view(np.uint16), tr[f"ln1_{li}"].view(np.uint16))
    # 2. qkv GEMM (+ RoPE + KV write): GEMM part within 1e-3; two rounded operands meet in RoPE: 2 ulps
    kc, vc = t(kv_before[li][0]), t(kv_before[li][1])
    qkv = e16(T, (nq + 2 * nkv) * d)
    if T <= 64:
        ops.qkv_rope_linear(t(tr[f"ln1_{li}"]), None, Lw.qkv_proj.weight, Lw.qkv_proj._scales(), qkv, inp["pos_t"],
                            model.cos_sin_cache, kc, vc, md.slot_mapping, nq, nkv, d)
    else:
        ops.w4a16_linear(t(tr[f"ln1_{li}"]), Lw.qkv_proj.weight, Lw.qkv_proj._scales(), qkv)
        ops.rope_kv_write(inp["pos_t"], qkv, model.cos_sin_cache, kc, vc, md.slot_mapping, nq, nkv, d)
    torch.cuda.synchronize()
    check("qkv v (GEMM only)", qkv[:, (nq + nkv) * d:].cpu().numpy(), tr[f"qkv_{li}"][:, (nq + nkv) * d:])
    check("qkv q,k (GEMM + RoPE)", qkv[:, :(nq + nkv) * d].cpu().numpy(), tr[f"qkv_{li}"][:, :(nq + nkv) * d], bar=2.5)
    check("key cache", kc.cpu().numpy(), kv_np[li][0], bar=2.5)
    check("value cache", vc.cpu().numpy(), kv_np[li][1])
    # 3. attention + head Hadamard on the oracle's q and the oracle's cache
    kc, vc = t(kv_np[li][0]), t(kv_np[li][1])
    ws = torch.zeros(ops.paged_attention_workspace_bytes(T, nq, d, md.n_splits), dtype=torch.uint8, device=DEV)
    had = e16(T, H)
    ops.paged_attention(t(tr[f"qkv_{li}"]), (nq + 2 * nkv) * d, kc, vc, md.block_tables, md.ctx_lens, md.q_start, T, q_len,
                        nq, d ** -0.5, md.n_splits, ws, None)
    ops.heads_hadamard_merged(ws, T, md.n_splits, T, nq, d, model.head_had_scale, out_f16=had)
    torch.cuda.synchronize()
    # 32 attention outputs (each within 1e-3) are mixed per Hadamard output: a few elements in 1e5 reach 2 ulps
    check("attention -> head Hadamard", had.cpu().numpy(), tr[f"o_in_{li}"], bar=2.5)
    # 4. o_proj on the oracle's Hadamard output
    o = e16(T, H)
    ops.w4a16_linear(t(tr[f"o_in_{li}"]), Lw.o_proj.weight, Lw.o_proj._scales(), o)
    torch.cuda.synchronize()
    check("o_proj", o.cpu().numpy(), tr[f"o_out_{li}"])
    # 5. residual add + post-attention norm: exact
    hid2 = e16(T, H)
    ops.add_rms_norm_fp16(normed, hid2, t(hid_in), t(tr[f"o_out_{li}"]), 1e-5)
    torch.cuda.synchronize()
    assert np.array_equal(hid2.cpu().numpy().view(np.uint16), tr[f"hidden_attn_{li}"].view(np.uint16))
    assert np.array_equal(normed.cpu().numpy().view(np.uint16), tr[f"ln2_{li}"].view(np.uint16))
    # 6. gate_up + silu*up: a product of two rounded GEMM outputs: 2 ulps
    act = e16(T, I)
    if T <= 64:
        ops.gate_up_silu_linear(t(tr[f"ln2_{li}"]), None, Lw.gate_up.weight, Lw.gate_up._scales(), act)
    else:
        gu = e16(T, 2 * I)
        ops.w4a16_linear(t(tr[f"ln2_{li}"]), Lw.gate_up.weight, Lw.gate_up._scales(), gu)
        ops.silu_mul(gu, act)
    torch.cuda.synchronize()
    check("gate_up -> silu*up", act.cpu().numpy(), tr[f"act_{li}"], bar=2.5)
    # 7. MLP Hadamard (fp16 out): exact
    hm = e16(T, I)
    ops.mlp_hadamard(t(tr[f"act_{li}"]), model.had_rem_dim, model.had_K, model.mlp_had_scale, out_f16=hm)
    torch.cuda.synchronize()
    assert np.array_equal(hm.cpu().numpy().view(np.uint16), tr[f"down_in_{li}"].view(np.uint16))
    # 8. down_proj on the oracle's Hadamard output (at T <= 16 as K slices finished inside the next norm)
    dn = e16(T, H)
    ops.w4a16_linear(t(tr[f"down_in_{li}"]), Lw.down_proj.weight, Lw.down_proj._scales(), dn)
    torch.cuda.synchronize()
    check("down_proj", dn.cpu().numpy(), tr[f"down_out_{li}"])
    S = ops.w4a16_linear_partial_slices(T, H, I) if T <= 16 else 0
    if 0 < S <= 4:
        part = torch.empty(S, T, H, dtype=torch.float32, device=DEV)
        ops.w4a16_linear_partial(t(tr[f"down_in_{li}"]), Lw.down_proj.weight, part, S)
        hid3 = e16(T, H)
        ops.add_rms_norm_fp16_partial(normed, hid3, t(tr[f"hidden_attn_{li}"]), part, Lw.down_proj._scales(), S, 1e-5)
        torch.cuda.synchronize()
        # hidden = residual + h(down): an absolute error of 1e-3 of the larger operand (the sum may cancel)
        a, b = tr[f"hidden_attn_{li}"].astype(np.float64), tr[f"down_out_{li}"].astype(np.float64)
        err = np.abs(hid3.cpu().numpy().astype(np.float64) - tr[f"hidden_{li}"].astype(np.float64))
        r = err / (1e-3 * np.maximum(1.0, np.maximum(np.abs(a), np.abs(b))))
        print(f"B={B} down_proj slices -> residual add in the norm: max err/1e-3 {r.max():.3f}")
        assert r.max() <= 1.0, r.max()


# ------------------------------------------------------------------ batch membership, capacity, the worker on the GPU

def _inject(eng, rng, B, k, V):
    U = rng.random((B, k)).astype(np.float32)
    E = rng.exponential(1.0, (B, k, V)).astype(np.float32)
    eng.inject_uniform, eng.inject_exponential = torch.from_numpy(U).to(DEV), torch.from_numpy(E).to(DEV)


def test_engine_slots_are_independent_and_empty_slots_are_inert(tiny):
    """A sequence's cycle does not depend on what the other slots hold: with the same injected draws, slots {0, 2} of a
    half-empty batch produce exactly the tokens they produce in a full batch; the empty slots write no KV, emit nothing
    and are not counted (the reference only ever sees scheduled sequences: spec_decode_base_sampler.py:127-129)."""
    from qspec_amd.spec_decode import QSpecEngine
    k, B, V = 3, 4, tiny.config.vocab_size
    prng = np.random.default_rng(21)
    prompts = [prng.integers(0, V, n).tolist() for n in (17, 33, 64, 5)]
    full = QSpecEngine(tiny, k, B, max_model_len=256, block_size=16, max_new_tokens=64, use_graph=False, seed=0)
    half = QSpecEngine(tiny, k, B, max_model_len=256, block_size=16, max_new_tokens=64, use_graph=False, seed=0)
    for b in range(B):                    # one prompt pass per request in both engines: the same KV history bit for
        full.add_sequence(b, prompts[b])  # bit (a batched prompt pass tiles its GEMMs differently: 1e-3-close, not identical)
    half.add_sequence(0, prompts[0])
    half.add_sequence(2, prompts[2])
    assert half.active_slots() == [0, 2]
    for cyc in range(3):
        rng = np.random.default_rng(100 + cyc)
        _inject(full, rng, B, k, V)
        half.inject_uniform, half.inject_exponential = full.inject_uniform, full.inject_exponential
        full.step(); half.step()
        torch.cuda.synchronize()
        of, oh = full.out_tokens.cpu(), half.out_tokens.cpu()
        assert torch.equal(of[[0, 2]], oh[[0, 2]]), cyc
        assert (oh[[1, 3]] == -1).all()
    assert torch.equal(full.seq_lens[[0, 2]], half.seq_lens[[0, 2]]) and half.seq_lens[[1, 3]].tolist() == [0, 0]
    for kc, vc in half.kv_caches:          # blocks of the empty slots: never written
        for b in (1, 3):
            rows = half.block_tables[b].long()
            assert not kc[rows].any() and not vc[rows].any()
    m = half.metrics()
    assert m.draft_tokens == 3 * 2 * k                       # two sequences, three cycles
    g_f, g_h = full.generated(), half.generated()
    assert g_f[0] == g_h[0] and g_f[2] == g_h[2] and g_h[1] == [] and g_h[3] == []


def test_engine_requests_join_and_leave_between_graph_replays(tiny):
    """The captured cycle serves a changing batch: a request leaves (free_slot), another joins (add_sequence) and one
    sits a step out (participants) between replays of the SAME graph; tokens equal the eager engine's."""
    from qspec_amd.spec_decode import QSpecEngine
    k, B, V = 3, 4, tiny.config.vocab_size
    prng = np.random.default_rng(22)
    prompts = [prng.integers(0, V, n).tolist() for n in (20, 9, 31, 12)]
    streams = []
    for use_graph in (False, True):
        eng = QSpecEngine(tiny, k, B, max_model_len=256, block_size=16, max_new_tokens=64, use_graph=use_graph, seed=7)
        eng.add_sequence(0, prompts[0]); eng.add_sequence(2, prompts[2])
        hist = {}
        eng.step(); eng.step()
        graph0 = eng._graph
        hist["req2"] = eng.generated()[2]
        eng.free_slot(2)                                  # request in slot 2 finished
        eng.add_sequence(3, prompts[3])                   # a new one joins in slot 3
        eng.step()
        eng.add_sequence(1, prompts[1])
        len0 = int(eng.seq_lens[0])
        eng.step(participants=[1, 3])                     # slot 0 sits this step out
        torch.cuda.synchronize()
        assert int(eng.seq_lens[0]) == len0
        eng.step()
        assert eng._graph is graph0                       # never re-captured
        gen = eng.generated()
        hist.update(req0=gen[0], req1=gen[1], req3=gen[3])
        assert gen[2] == []
        streams.append(hist)
        eng.sync_lens()
        assert eng._len_ub == eng.seq_lens.tolist()
    assert streams[0] == streams[1]
    assert all(len(v) >= 2 for v in streams[0].values())


def test_engine_capacity_guard_refuses_cleanly(tiny):
    """A cycle that could write beyond a sequence's block table / max_model_len is refused on the host before anything is
    enqueued (ValueError), and the kernels themselves drop such a write (slot -1) instead of walking off the block-table
    row into the neighbour's blocks."""
    from qspec_amd import ops
    from qspec_amd.spec_decode import QSpecEngine
    k, B, V = 3, 2, tiny.config.vocab_size
    prng = np.random.default_rng(23)
    eng = QSpecEngine(tiny, k, B, max_model_len=48, block_size=16, max_new_tokens=64, use_graph=True, seed=1)
    eng.add_sequence(0, prng.integers(0, V, 30).tolist())
    eng.add_sequence(1, prng.integers(0, V, 8).tolist())
    with pytest.raises(ValueError):
        eng.add_sequence(1, [1, 2, 3])                    # occupied
    with pytest.raises(ValueError, match="max_position_embeddings"):
        QSpecEngine(tiny, k, B, max_model_len=tiny.config.max_position_embeddings + 16)   # beyond the RoPE table
    n = 0
    with pytest.raises(ValueError, match="slot 0"):
        for _ in range(20):
            eng.step()
            eng.sync_lens()                               # exact lengths, as the worker has them
            n += 1
    assert 3 <= n <= 16                                   # 1 .. k+1 tokens per cycle from length 31 to the 48-token limit
    torch.cuda.synchronize()
    lens = eng.seq_lens.tolist()
    assert lens[0] + k <= 48 + k and lens[0] <= 48        # nothing ran past the guard
    snap = [(kc.clone(), vc.clone()) for kc, vc in eng.kv_caches]
    with pytest.raises(ValueError):
        eng.step()                                        # still refused, still nothing enqueued
    torch.cuda.synchronize()
    assert eng.seq_lens.tolist() == lens
    for (kc, vc), (k0, v0) in zip(eng.kv_caches, snap):
        assert torch.equal(kc, k0) and torch.equal(vc, v0)
    # the in-kernel safety net: a length beyond the row's blocks gives slot -1, not the next row's block
    bt = eng.block_tables
    seq_lens = torch.tensor([bt.shape[1] * 16 + 5, 10], dtype=torch.int32, device=DEV)
    last = torch.zeros(2, dtype=torch.int64, device=DEV)
    tok = torch.zeros(2, dtype=torch.int64, device=DEV); pos = torch.zeros_like(tok); slots = torch.zeros_like(tok)
    ctx = torch.zeros(2, dtype=torch.int32, device=DEV)
    ops.spec_prepare_draft(last, seq_lens, bt, 16, tok, pos, slots, ctx)
    torch.cuda.synchronize()
    assert slots.tolist()[0] == -1 and slots.tolist()[1] == int(bt[1, 0]) * 16 + 9
    eng.free_slot(0)                                      # the long request finishes: the other one carries on
    eng.step()
    torch.cuda.synchronize()
    assert eng.out_tokens[0].tolist() == [-1] * (k + 1) and int(eng.out_tokens[1, 0]) != -1


def test_worker_variable_batch_on_the_gpu(tiny):
    """execute_model with a changing seq_group_metadata_list: prompts admitted while others decode, a finished request
    leaves, order shuffled; every output row belongs to the request named beside it and the worker's view of the
    lengths equals the GPU's after every call (spec_decode_worker.py:461-560,972-1063)."""
    from qspec_amd.spec_decode import ExecuteModelRequest, SequenceGroupMetadata, create_spec_worker
    from qspec_amd.spec_decode.worker import SequenceData, SpeculativeConfig
    rng = np.random.default_rng(24)
    w = create_spec_worker(model_config=tiny.config, model=tiny, speculative_config=SpeculativeConfig(3),
                           max_num_seqs=4, max_model_len=256, block_size=16, device=DEV)
    w.init_device()
    w.initialize_cache(64, 0)
    eng = w.engine
    toks = {}
    rid_of = {}

    def mk(rid, n):
        rid_of[ord(rid)] = rid
        return SequenceGroupMetadata(rid, True, {ord(rid): SequenceData(rng.integers(0, 2048, n).tolist())})

    def absorb(outs):
        for o in outs:
            for grp in o.outputs:                       # request order; the sequence id names the request
                smp = grp.samples[0]
                if smp.output_token != -1:
                    toks.setdefault(rid_of[smp.parent_seq_id], []).append(smp.output_token)
        torch.cuda.synchronize()
        assert eng._len_ub == eng.seq_lens.tolist()

    a, b, c = mk("a", 12), mk("b", 30), mk("c", 7)
    absorb(w.execute_model(ExecuteModelRequest([a, b], num_lookahead_slots=0)))
    for s in (a, b):
        s.is_prompt = False
    absorb(w.execute_model(ExecuteModelRequest([b, a], num_lookahead_slots=3)))
    absorb(w.execute_model(ExecuteModelRequest([c], num_lookahead_slots=0)))                  # joins while a, b wait
    c.is_prompt = False
    absorb(w.execute_model(ExecuteModelRequest([a, c, b], num_lookahead_slots=3)))
    absorb(w.execute_model(ExecuteModelRequest([c, b], num_lookahead_slots=3, finished_requests_ids=["a"])))
    assert "a" not in w._slots and sorted(w._slots) == ["b", "c"]
    d = mk("d", 20)
    absorb(w.execute_model(ExecuteModelRequest([d], num_lookahead_slots=0)))                  # takes a's slot
    d.is_prompt = False
    assert w._slots["d"] == 0
    absorb(w.execute_model(ExecuteModelRequest([d, b, c], num_lookahead_slots=3)))
    # the tokens reported per request are the tokens the engine holds for the request's slot
    gen = eng.generated()
    for rid in ("b", "c", "d"):
        assert toks[rid] == gen[w._slots[rid]], rid
    assert len(toks["a"]) >= 3
    assert w.execute_model(None) == []


@pytest.mark.parametrize("k,batch_size", [(1, 1), (2, 2), (6, 4)])
def test_correctly_calls_spec_decode_sampler(tiny, k, batch_size):
    """tests/spec_decode/test_spec_decode_worker.py:148-233 (`test_correctly_calls_spec_decode_sampler`) on the real
    engine: the rejection sampler must be called ONCE per cycle with target_with_bonus_probs = the scorer's [B, k+1, V]
    distributions, bonus_token_ids = the scorer's token at the last position, draft_probs / draft_token_ids = the
    proposer's k steps in [B, k, ...] order -- checked as the reference test does, by stopping the step inside the sampler."""
    from qspec_amd.spec_decode import QSpecEngine
    rng = np.random.default_rng(k * 10 + batch_size)
    eng = QSpecEngine(tiny, k, batch_size, max_model_len=128, block_size=16, max_new_tokens=32, use_graph=False, seed=1)
    eng.add_sequences([rng.integers(0, tiny.config.vocab_size, 9 + 3 * b).tolist() for b in range(batch_size)])
    calls = []
    secret = "artificial stop"

    def spy(target_with_bonus_probs, bonus_token_ids, draft_probs, draft_token_ids, seeded_seqs=None, **kw):
        calls.append(SimpleNamespace(target_with_bonus_probs=target_with_bonus_probs, bonus_token_ids=bonus_token_ids,
                                     draft_probs=draft_probs, draft_token_ids=draft_token_ids, kw=kw))
        raise ValueError(secret)
    from types import SimpleNamespace
    eng.sampler.forward = spy
    with pytest.raises(ValueError, match=secret):
        eng.step()
    torch.cuda.synchronize()
    assert len(calls) == 1
    a = calls[0]
    V = tiny.config.vocab_size
    assert a.target_with_bonus_probs.shape == (batch_size, k + 1, V) and a.target_with_bonus_probs.data_ptr() == eng.target_probs.data_ptr()
    assert torch.equal(a.bonus_token_ids.reshape(batch_size, 1), eng.target_tokens[:, -1:])
    assert a.draft_probs.shape == (batch_size, k, V) and torch.equal(a.draft_probs, eng.draft_probs_kbv.transpose(0, 1))
    assert a.draft_token_ids.shape == (batch_size, k) and torch.equal(a.draft_token_ids, eng.draft_ids_kb.t())
    # and the distributions are distributions, the proposals their argmax (greedy draft, sampler.py:270-287)
    assert torch.allclose(a.draft_probs.sum(-1), torch.ones(batch_size, k, device=DEV), atol=1e-4)
    assert torch.equal(a.draft_probs.argmax(-1), a.draft_token_ids)
    assert torch.equal(a.target_with_bonus_probs[:, -1].argmax(-1), a.bonus_token_ids.reshape(-1))


@pytest.mark.parametrize("width,B", [("tiny", 4), ("llama-3-8b", 4), ("llama-3-8b", 8)])
def test_cycle_recovery_replay_is_bit_identical(tiny, width, B):
    """A cycle whose error word is non-zero (a device-side hand-off timed out) is re-run from the state snapshot taken at
    its start, without hand-offs (engine.recover): the replay must emit exactly what an undisturbed engine emits --
    sequence state, sampler counters and the Philox state restored, the one-workgroup kernel forms bit-identical.  At the
    Llama-3-8B width the verify pass runs on fragment-major tiles (B = 4: 16 tokens) / the two-token-tile kernel (B = 8: 32
    tokens): the replay must take the same GEMM forms (the MLP transform's one-workgroup form stores the tiles too)."""
    from qspec_amd import ops
    from qspec_amd.model import QuarotLlamaConfig, QuarotLlamaForCausalLM
    from qspec_amd.spec_decode import QSpecEngine
    rng = np.random.default_rng(14)
    if width != "tiny":
        cfg = QuarotLlamaConfig(4096, 14336, 32, 8, 2, 1024, 1e-5, 500000.0, 512, "llama-3-8b-2layer")
        tiny = QuarotLlamaForCausalLM(cfg, DEV).init_synthetic(seed=4, lm_head_std=0.05)
    prompts = [rng.integers(0, tiny.config.vocab_size, n).tolist() for n in (20, 31, 8, 50, 13, 27, 9, 40)[:B]]
    runs = []
    for poke in (False, True):
        eng = QSpecEngine(tiny, 3, B, max_model_len=256, block_size=16, max_new_tokens=64, use_graph=False, seed=11)
        eng.add_sequences(prompts)
        eng.step()
        if width != "tiny":
            assert tiny.last_forward_form == ("fragment-major" if B == 4 else "two-tile")
        out0, err = eng.read_outputs()
        assert err == 0
        eng.note_emitted([int((out0[b] != -1).sum()) for b in range(B)])
        if poke:   # raise this stream's sticky exchange-workspace error word: the next cycle reports it
            ops.xwg_workspace(DEV)[:1].fill_(1)
        eng.step()
        out, err = eng.read_outputs()
        assert (err != 0) == poke
        if poke:
            eng.recover()
            out, err = eng.read_outputs()
            assert err == 0 and eng.recoveries == 1
        eng.note_emitted([int((out[b] != -1).sum()) for b in range(B)])
        eng.step()                                                # and the engine carries on from the recovered state
        out2, err2 = eng.read_outputs()
        assert err2 == 0
        runs.append((out.clone(), out2.clone(), eng.generated(), eng.metrics(), eng.sampler.rng_state.tolist()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    assert runs[0][2:] == runs[1][2:]


def test_fallback_mode_reports_and_recovers_errors(tiny):
    """The draft-graph + eager-verify fallback (taken when the cycle's collectives cannot be captured) carries the same
    bracket as the captured cycle: the state snapshot at the head of the draft graph, the error words collected behind the
    eager verify pass.  Forced here by failing the whole-cycle capture once; a poked error word must surface through
    read_outputs() and the recovery replay must emit what an undisturbed engine emits."""
    import warnings
    from qspec_amd import ops
    from qspec_amd.spec_decode import QSpecEngine

    class FailsWholeCapture(QSpecEngine):
        fail_once = True

        def _collect_errors(self):
            if self.fail_once and torch.cuda.is_current_stream_capturing():
                type(self).fail_once = False
                raise RuntimeError("forced: this communicator cannot be captured")
            super()._collect_errors()
    rng = np.random.default_rng(15)
    prompts = [rng.integers(0, tiny.config.vocab_size, n).tolist() for n in (12, 25, 7, 40)]
    runs = []
    for cls, poke in ((QSpecEngine, False), (FailsWholeCapture, True)):
        eng = cls(tiny, 3, 4, max_model_len=256, block_size=16, max_new_tokens=64, use_graph=True, seed=12)
        eng.add_sequences(prompts)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            eng.step()
        if poke:
            assert eng._graph is None and eng._graph_draft is not None, "the fallback mode was not taken"
        out0, err = eng.read_outputs()
        assert err == 0
        eng.note_emitted([int((out0[b] != -1).sum()) for b in range(4)])
        if poke:
            ops.xwg_workspace(DEV)[:1].fill_(1)
        eng.step()
        out, err = eng.read_outputs()
        assert (err != 0) == poke
        if poke:
            eng.recover()
            out, err = eng.read_outputs()
            assert err == 0 and eng.recoveries == 1
        eng.note_emitted([int((out[b] != -1).sum()) for b in range(4)])
        eng.step()
        out2, err2 = eng.read_outputs()
        assert err2 == 0
        runs.append((out0.clone(), out.clone(), out2.clone(), eng.generated(), eng.metrics()))
    for a, b in zip(runs[0][:3], runs[1][:3]):
        assert torch.equal(a, b)
    assert runs[0][3:] == runs[1][3:]


def test_worker_from_vllm_config_loads_a_checkpoint_directory(tiny, tmp_path):
    """create_spec_worker(vllm_config=...) -> init_device loads `model_config.model` (a directory in the reference's
    on-disk format) through qspec_amd/checkpoint.py, and the worker then emits what the in-memory model emits."""
    from types import SimpleNamespace
    from qspec_amd import checkpoint
    from qspec_amd.spec_decode import ExecuteModelRequest, SamplingParams, SequenceData, SequenceGroupMetadata, create_spec_worker
    checkpoint.save_qspec_checkpoint(tiny, str(tmp_path))
    c = tiny.config
    hf = SimpleNamespace(model_type="llama_quarot", hidden_size=c.hidden_size, intermediate_size=c.intermediate_size,
                         num_attention_heads=c.num_attention_heads, num_key_value_heads=c.num_key_value_heads,
                         num_hidden_layers=c.num_hidden_layers, vocab_size=c.vocab_size, rms_norm_eps=c.rms_norm_eps,
                         rope_theta=c.rope_theta, max_position_embeddings=c.max_position_embeddings)
    cfg = SimpleNamespace(
        model_config=SimpleNamespace(hf_config=hf, max_model_len=256, model=str(tmp_path), seed=0, max_logprobs=5),
        cache_config=SimpleNamespace(block_size=16, gpu_memory_utilization=0.9, swap_space_bytes=0),
        scheduler_config=SimpleNamespace(max_num_seqs=2, max_num_batched_tokens=512),
        parallel_config=SimpleNamespace(tensor_parallel_size=1, pipeline_parallel_size=1),
        load_config=SimpleNamespace(load_format="auto"),
        speculative_config=SimpleNamespace(num_speculative_tokens=3, speculative_disable_by_batch_size=None,
                                           disable_log_stats=False, disable_logprobs=False,
                                           draft_token_acceptance_method="rejection_sampler"))
    rng = np.random.default_rng(6)
    prompts = [rng.integers(0, c.vocab_size, n).tolist() for n in (9, 14)]
    results = []
    for kw in (dict(vllm_config=cfg, local_rank=0, rank=0, distributed_init_method=None, is_driver_worker=True),
               dict(vllm_config=cfg, local_rank=0, rank=0, model=tiny)):
        w = create_spec_worker(**kw)
        w.init_device()
        w.load_model()
        assert w.get_model() is not None and (w.get_model() is tiny) == ("model" in kw)
        w.initialize_cache(32, 0)
        sp = SamplingParams(logprobs=2)
        sgs = [SequenceGroupMetadata(f"r{i}", True, {i: SequenceData(p)}, sampling_params=sp) for i, p in enumerate(prompts)]
        out = w.execute_model(ExecuteModelRequest(sgs, num_lookahead_slots=0))
        seq = [out[0].token_ids()]
        dec = [SequenceGroupMetadata(s.request_id, False, s.seq_data, sampling_params=sp) for s in sgs]
        outs = w.execute_model(ExecuteModelRequest(dec, num_lookahead_slots=3))
        seq += [o.token_ids() for o in outs]
        # disable_logprobs=False: real target logprobs -- the sampled token's entry + the two best, ranks from 1
        lp = outs[0].outputs[0].samples[0].logprobs
        tok = outs[0].outputs[0].samples[0].output_token
        assert tok in lp and lp[tok].rank >= 1 and lp[tok].logprob <= 0.0 and 2 <= len(lp) <= 3
        assert 1 in [v.rank for v in lp.values()]
        assert outs[0].spec_decode_worker_metrics is None or outs[0].spec_decode_worker_metrics.num_spec_tokens == 3
        results.append(seq)
    assert results[0] == results[1]


# ------------------------------------------------------------------ checkpoint on disk -> HIP path

def test_reference_format_checkpoint_runs_on_the_hip_path(oracle, tmp_path):
    """A checkpoint written the way the reference's exporter writes it -- per projection `int = clamp(round(W / scale),
    -8, 7)`, `pack_i4`, `weight_scales` [N, 1], Sequential indices in the o_proj / down_proj names, separate q / k / v
    and up / gate tensors, two safetensors shards (third-party/QuaRot/e2e/checkpoint_utils/quantize_llama_checkpoint.py:
    87-119) -- is read by the loader (vllm/worker/model_runner.py:1132-1148 + fuse_qkv / fuse_gate_up,
    quarot_llama.py:152-173,301-314) and runs both passes on the GPU; the oracle consumes the SAME raw tensors with the
    fusion done here in numpy, so a wrong row order / rename / nibble convention in the loader cannot cancel out.
    (No reference checkpoint is reachable offline: the files are made here; parity of the file format is unpinned.)"""
    from safetensors.torch import save_file
    from oracle.model import OracleModel
    from qspec_amd import checkpoint, hadamard_tables
    from qspec_amd.model import QuarotLlamaForCausalLM, Scratch
    cfg = tiny_cfg()
    H, I, L, V = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.vocab_size
    q_sz, kv_sz = cfg.q_size, cfg.kv_size
    rng = np.random.default_rng(33)
    had28 = hadamard_tables.get_hadK(I)[0].to(torch.float16)
    disk, raw = {}, []

    def export(name, n, kdim):   # the exporter's arithmetic on an fp16 weight with a per-channel scale
        w = (rng.standard_normal((n, kdim)) * 0.05).astype(np.float16)
        scale = (np.abs(w.astype(np.float32)).max(axis=1, keepdims=True) / 7.0).astype(np.float16)
        q = np.clip(np.round(w.astype(np.float32) / scale.astype(np.float32)), -8, 7).astype(np.int8)
        packed = oracle.pack_i4(q)
        disk[f"{name}.weight"] = torch.from_numpy(packed.view(np.uint8).copy())
        disk[f"{name}.weight_scales"] = torch.from_numpy(scale.copy())
        return packed, scale.reshape(-1)

    for i in range(L):
        p = f"model.layers.{i}"
        qw, kw, vw = (export(f"{p}.self_attn.{n}_proj", sz, H) for n, sz in (("q", q_sz), ("k", kv_sz), ("v", kv_sz)))
        ow = export(f"{p}.self_attn.o_proj.1", H, H)
        uw, gw = export(f"{p}.mlp.up_proj", I, H), export(f"{p}.mlp.gate_proj", I, H)
        dw = export(f"{p}.mlp.down_proj.2", H, I)
        disk[f"{p}.mlp.down_proj.0.had_rem_dim"] = had28.clone()
        raw.append(dict(qkv_w=np.concatenate([qw[0], kw[0], vw[0]]), qkv_s=np.concatenate([qw[1], kw[1], vw[1]]),
                        o_w=ow[0], o_s=ow[1], gate_up_w=np.concatenate([uw[0], gw[0]]),
                        gate_up_s=np.concatenate([uw[1], gw[1]]), down_w=dw[0], down_s=dw[1]))
    embed = (rng.standard_normal((V, H)) * 0.02).astype(np.float16)
    head = (rng.standard_normal((V, H)) * 0.05).astype(np.float16)
    disk["model.embed_tokens.weight"], disk["lm_head.weight"] = torch.from_numpy(embed.copy()), torch.from_numpy(head.copy())
    keys = sorted(disk)
    save_file({k: disk[k] for k in keys[::2]}, str(tmp_path / "model-00001-of-00002.safetensors"))
    save_file({k: disk[k] for k in keys[1::2]}, str(tmp_path / "model-00002-of-00002.safetensors"))

    model = QuarotLlamaForCausalLM(cfg, DEV)
    checkpoint.load_qspec_checkpoint(model, str(tmp_path))
    om = OracleModel(cfg, raw, embed, head, had28.numpy(), model.had_K, model.cos_sin_cache.cpu().numpy(), 16)
    for w4a4, q_len in ((True, 1), (False, 3)):
        irng = np.random.default_rng(5)
        inp = make_inputs(model, irng, [40, 130, 9], q_len)
        kv_np = [(k.copy(), v.copy()) for k, v in inp["kv_np"]]
        ref = om.forward(inp["ids"], inp["pos"], kv_np, inp["slots"], inp["bt"], inp["ctx"], inp["q_start"], w4a4)
        s = Scratch(cfg, inp["T"], 3, q_len, inp["n_splits"], DEV)
        out = model.forward(inp["ids_t"], inp["pos_t"], inp["kv_t"], inp["md"], s, w4a4=w4a4)
        torch.cuda.synchronize()
        diff = np.abs(out.cpu().numpy().astype(np.float64) - ref.astype(np.float64))
        if w4a4:   # first layer up to attention: bit-exact KV (norm, int4 GEMM over the loaded q/k/v rows, RoPE)
            assert np.array_equal(inp["kv_t"][0][0].cpu().numpy().view(np.uint16), kv_np[0][0].view(np.uint16))
            assert np.array_equal(inp["kv_t"][0][1].cpu().numpy().view(np.uint16), kv_np[0][1].view(np.uint16))
            assert np.median(diff) < 1e-3 and np.quantile(diff, 0.99) < 0.1, (np.median(diff), np.quantile(diff, 0.99))
        else:
            assert diff.max() < 2e-2 and np.median(diff) < 1e-3, (np.median(diff), diff.max())
        logits = model.compute_logits(out, s).cpu().numpy().astype(np.float64)
        assert np.quantile(np.abs(logits - om.logits(ref).astype(np.float64)), 0.99) < (0.1 if w4a4 else 3e-2)


def test_structured_weights_draft_tracks_target():
    """Acceptance earned, not injected: a model whose next-token distribution is well conditioned -- tied lm_head =
    embedding (each token predicts itself with a wide margin), decoder layers acting as a ~10 % perturbation of the
    residual stream -- makes the W4A4 draft and the W4A16 target agree, and the rejection sampler accepts nearly
    every proposal (the reference reports 0.961 on its trained checkpoint, figs/image-1.png).  With the random weights
    of the benchmarks the same code accepts ~1-4 %: that is a property of random weights, not of the draft path."""
    from qspec_amd.model import QuarotLlamaForCausalLM
    from qspec_amd.spec_decode import QSpecEngine
    cfg = tiny_cfg()
    model = QuarotLlamaForCausalLM(cfg, DEV).init_synthetic(seed=11, lm_head_std=0.02)
    with torch.no_grad():
        for layer in model.layers:
            for lin in layer.linears():
                lin.weight_scales.mul_(2e-3)
        model.lm_head.copy_(model.embed_tokens)
    k, B, V = 3, 4, cfg.vocab_size
    rng = np.random.default_rng(12)
    eng = QSpecEngine(model, k, B, max_model_len=256, block_size=16, max_new_tokens=128, use_graph=True, seed=3)
    eng.add_sequences([rng.integers(0, V, n).tolist() for n in (9, 20, 33, 14)])
    for _ in range(10):
        eng.step()
    m = eng.metrics()
    assert m.draft_tokens == 10 * B * k
    assert m.draft_acceptance_rate > 0.9 and m.system_efficiency > 0.85, m
    # the draft's distribution is close to the target's where it matters: total variation at the verified positions
    tv = 0.5 * (eng.draft_probs_kbv.transpose(0, 1) - eng.target_probs[:, :k]).abs().sum(-1)
    assert float(tv.max()) < 0.1, tv
