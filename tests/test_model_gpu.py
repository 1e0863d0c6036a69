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
        assert diff.max() < 2e-2 and np.median(diff) < 1e-3, (np.median(diff), diff.max())
    # layer-0 KV written by the HIP path equals the oracle's exactly for W4A4 (no attention upstream of it)
    if w4a4:
        k_hip = inp["kv_t"][0][0].cpu().numpy()
        assert np.array_equal(k_hip.view(np.uint16), kv_np[0][0].view(np.uint16))
    logits = tiny.compute_logits(out, s).cpu().numpy().astype(np.float64)
    ref_logits = om.logits(ref).astype(np.float64)
    assert np.quantile(np.abs(logits - ref_logits), 0.99) < (0.1 if w4a4 else 3e-2)


def test_engine_cycle_matches_oracle_engine(tiny, oracle):
    """Three full cycles (k=3, B=4).  Two HIP-vs-oracle differences are legitimate: fp32 summation order inside
    attention / the W4A16 MFMA, and (a consequence) a rare different int4 value downstream.  So:
      * numerics: the oracle engine is teacher-forced with the GPU's draft tokens; its draft / target
        distributions must be close to the GPU's (total-variation distance per row);
      * logic: everything discrete -- accept mask, recovered ids, output layout, counters, sequence state,
        KV slot bookkeeping -- must be EXACT given the GPU's own distributions and the injected draws."""
    from oracle.model import OracleEngine, OracleModel
    from qspec_amd.spec_decode import QSpecEngine
    rng = np.random.default_rng(2)
    k, B, V = 3, 4, tiny.config.vocab_size
    prompts = [rng.integers(0, V, n).tolist() for n in (17, 33, 64, 5)]
    eng = QSpecEngine(tiny, k, B, max_model_len=256, block_size=16, max_new_tokens=64, use_graph=False, seed=0)
    eng.add_sequences(prompts)
    oe = OracleEngine(OracleModel.from_torch_model(tiny, 16), k, B, 256, 16)
    oe.add_sequences(prompts)
    assert eng.last_token.tolist() == oe.last_token.tolist()      # prefill: greedy target token
    gen = [[int(t)] for t in eng.last_token.tolist()]
    counters = np.zeros(3, np.int64)
    all_tv_d, all_tv_t = [], []
    for cyc in range(3):
        U = rng.random((B, k)).astype(np.float32)
        E = rng.exponential(1.0, (B, k, V)).astype(np.float32)
        eng.inject_uniform, eng.inject_exponential = torch.from_numpy(U).to(DEV), torch.from_numpy(E).to(DEV)
        L0 = eng.seq_lens.cpu().numpy().copy()
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
    tv_d, tv_t = np.concatenate(all_tv_d).ravel(), np.concatenate(all_tv_t).ravel()
    # (measured: a 1e-7 attention difference flips an fp16 ulp in ~20% of rows, the int4 pipeline amplifies it to
    # a total-variation distance of ~0.05-0.08 after two layers)
    assert np.median(tv_d) < 0.12 and tv_d.max() < 0.6 and tv_d.min() < 1e-3, tv_d
    assert np.median(tv_t) < 5e-3 and tv_t.max() < 0.15, tv_t
    assert eng.generated() == gen
    m = eng.metrics()
    assert [m.accepted_tokens, m.emitted_tokens, m.draft_tokens] == counters.tolist()
    rate, eff = oracle.spec_metrics(*counters.tolist(), k)
    assert abs(m.draft_acceptance_rate - rate) < 1e-12 and abs(m.system_efficiency - eff) < 1e-12


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
    nb, _ = w.determine_num_available_blocks()
    w.initialize_cache(nb, 0)
    with pytest.raises(NotImplementedError):
        w.get_cache_block_size_bytes()
    sg = [SequenceGroupMetadata(f"r{i}", True, {i: SequenceData(rng.integers(0, 2048, 10 + i).tolist())}) for i in range(4)]
    out = w.execute_model(ExecuteModelRequest(sg, num_lookahead_slots=0))
    assert len(out) == 1 and out[0].sampled_token_ids.shape == (4,) and (out[0].sampled_token_ids >= 0).all()
    for s in sg:
        s.is_prompt = False
    req = ExecuteModelRequest(sg, num_lookahead_slots=3)
    outs = w.execute_model(req)
    assert req.w4a4 is False and w.proposer_calls == 3 and w.scorer_calls == 2
    assert 1 <= len(outs) <= 4
    toks = torch.stack([o.sampled_token_ids for o in outs], 1)    # [B, steps]
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
        assert diff.max() < 2e-2 and np.median(diff) < 1e-3, (np.median(diff), diff.max())
    # and the reference-order module-wise path gives the same bits as the fused one
    kv_b = [(torch.from_numpy(k).to(DEV), torch.from_numpy(v).to(DEV)) for k, v in inp["kv_np"]]
    b = model.forward_modulewise(inp["ids_t"], inp["pos_t"], kv_b, inp["md"], w4a4=w4a4)
    torch.cuda.synchronize()
    if w4a4:
        assert torch.equal(out.view(torch.int16), b.view(torch.int16))
    else:
        assert (out.float() - b.float()).abs().max().item() < 2e-2


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
    nb, _ = w.determine_num_available_blocks()
    w.initialize_cache(nb, 0)
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
    assert len(outs) == 1 and torch.equal(outs[0].sampled_token_ids, expect.cpu())
    assert (w.proposer_calls, w.scorer_calls) == (calls[0], calls[1] + 1)
    assert torch.equal(eng.seq_lens, seq0 + 1) and torch.equal(eng.last_token, expect)
    outs = w.execute_model(ExecuteModelRequest(sg, num_lookahead_slots=3, running_queue_size=2))
    assert len(outs) == 1 and (outs[0].sampled_token_ids >= 0).all() and w.proposer_calls == calls[0]
    # and a speculative step still works afterwards
    outs = w.execute_model(ExecuteModelRequest(sg, num_lookahead_slots=3))
    assert 1 <= len(outs) <= 4 and w.proposer_calls == calls[0] + 3
