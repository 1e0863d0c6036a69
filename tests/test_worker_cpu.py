"""SpecDecodeWorker host logic on the CPU with a scripted engine (no GPU, no model): request -> slot mapping, order
and subset handling, finished requests, output formatting, bonus-token bookkeeping, the speculation-off paths and the
tensor-parallel control plane.  Scenarios follow the reference's mock-based worker tests
(tests/spec_decode/test_spec_decode_worker.py:148-370 output format / sampler wiring, :472-570 k = 0 and empty batch,
:687-823 bonus tokens and finished requests) against vllm/spec_decode/spec_decode_worker.py:461-560,722-755,972-1063,
1178-1210."""
import threading

import pytest
import torch

from qspec_amd.model import QuarotLlamaConfig
from types import SimpleNamespace

from qspec_amd.spec_decode.worker import (CompletionSequenceGroupOutput, DeviceHandoffTimeout, ExecuteModelRequest,
                                          Logprob, SamplingParams, SequenceData, SequenceGroupMetadata, SequenceOutput,
                                          SpeculativeConfig, create_spec_worker)


class FakeSampler:
    def __init__(self):
        self.counters = torch.zeros(3, dtype=torch.long)
        self.num_accepted_tokens = self.counters[0]
        self.num_emitted_tokens = self.counters[1]
        self.num_draft_tokens = 0


class FakeEngine:
    """The QSpecEngine surface the worker uses, with scripted outputs: out_script(step_index, slots) -> [B, k+1]."""

    def __init__(self, model, k, B, max_model_len, block_size, seed=0, num_blocks=None, acceptance_sampler=None):
        self.k, self.B = k, B
        self.acceptance_sampler = acceptance_sampler
        self.num_blocks = num_blocks
        self.err_script = []          # error words the next read_outputs() calls return (then 0)
        self.recoveries = 0
        self.sampler = FakeSampler()
        self.out_tokens = torch.full((B, k + 1), -1, dtype=torch.int64)
        self.gen_tokens = torch.full((B, 64), -1, dtype=torch.int64)
        self._len_ub = [0] * B
        self._len_before = [0] * B
        self.block_tables = {}
        self.calls = []
        self.out_script = None
        self.steps = 0

    def add_sequence(self, slot, prompt, block_table=None, sync=True):
        assert self._len_ub[slot] == 0
        self.calls.append(("add", slot, len(prompt), tuple(block_table) if block_table is not None else None))
        self._len_ub[slot] = len(prompt) + 1
        self.gen_tokens[slot, 0] = 1000 + slot

    def add_sequences_to(self, slots, prompts, block_tables=None):
        for i, (b, p) in enumerate(zip(slots, prompts)):
            self.add_sequence(b, p, None if block_tables is None else block_tables[i])

    def free_slot(self, slot):
        self.calls.append(("free", slot))
        self._len_ub[slot] = 0

    def set_block_table(self, slot, blocks):
        self.block_tables[slot] = list(blocks)

    def set_sampling_params(self, slot, temperature=0.0, top_k=-1, top_p=1.0):
        self.sampling = getattr(self, "sampling", {})
        self.sampling[slot] = (temperature, top_k, top_p)

    def _script(self, slots):
        self.out_tokens.fill_(-1)
        out = self.out_script(self.steps, slots) if self.out_script else None
        self.steps += 1
        return out

    def step(self, participants=None):
        slots = list(participants)
        self.calls.append(("step", tuple(slots)))
        self._len_before = list(self._len_ub)
        out = self._script(slots)
        for b in slots:
            row = out[b] if out is not None else [7] + [-1] * self.k
            self.out_tokens[b] = torch.tensor(row, dtype=torch.int64)

    def step_no_spec(self, participants=None):
        slots = list(participants)
        self.calls.append(("step_no_spec", tuple(slots)))
        self._len_before = list(self._len_ub)
        self._script(slots)
        for b in slots:
            self.out_tokens[b, 0] = 500 + b

    def note_emitted(self, emitted):
        for b, n in enumerate(emitted):
            if self._len_ub[b] > 0:
                self._len_ub[b] = self._len_before[b] + n

    def error_flag(self):
        return 0

    def read_outputs(self):
        return self.out_tokens.clone(), (self.err_script.pop(0) if self.err_script else 0)

    def recover(self):
        self.calls.append(("recover",))
        self.recoveries += 1


CFG = QuarotLlamaConfig(1024, 3584, 8, 2, 2, 2048, 1e-5, 10000.0, 512, "tiny")


GiB = 1 << 30


def fake_memory_probe():
    """(free bytes with the weights loaded, total bytes, torch peak increase of the profiled prompt pass + cycle)."""
    return 200 * GiB, 288 * GiB, 2 * GiB


def make_worker(k=3, B=4, disable_by_batch_size=None, model=None, rank=0, **kw):
    w = create_spec_worker(model_config=CFG, model=model if model is not None else object(),
                           speculative_config=SpeculativeConfig(k, speculative_disable_by_batch_size=disable_by_batch_size),
                           max_num_seqs=B, max_model_len=256, block_size=16, device="cpu", engine_factory=FakeEngine,
                           disable_log_stats=True, rank=rank, memory_probe=fake_memory_probe, **kw)
    w.init_device()
    nb, _ = w.determine_num_available_blocks()
    w.initialize_cache(nb, 0)
    return w


def prompt(rid, seq_id, n, blocks=None):
    return SequenceGroupMetadata(rid, True, {seq_id: SequenceData(list(range(n)))},
                                 block_tables={seq_id: blocks} if blocks is not None else None)


def decode(sg):
    return SequenceGroupMetadata(sg.request_id, False, sg.seq_data, block_tables=sg.block_tables)


def test_prefill_assigns_slots_and_honours_block_tables():
    w = make_worker()
    sg = [prompt("a", 10, 5, [3, 4]), prompt("b", 11, 9)]
    out = w.execute_model(ExecuteModelRequest(sg, num_lookahead_slots=0))
    assert len(out) == 1 and [o.samples[0].parent_seq_id for o in out[0].outputs] == [10, 11]
    assert out[0].token_ids() == [1000, 1001]                         # the target's first token of each prompt
    assert w.engine.calls == [("add", 0, 5, (3, 4)), ("add", 1, 9, None)]
    assert w.proposer_calls == 0 and w.scorer_calls == 1              # the proposer never runs on prefill (:699)
    with pytest.raises(AssertionError):
        w.execute_model(ExecuteModelRequest([prompt("c", 12, 4)], num_lookahead_slots=3))   # prompt-only => 0 slots


def test_output_format_order_and_subset():
    """Requests may come in any order and any subset; one SamplerOutput per emitted position, -1 = nothing, trailing
    all-(-1) steps dropped (:972-1063)."""
    k = 3
    w = make_worker(k=k)
    sgs = [prompt(r, i, 4 + i) for i, r in enumerate("abcd")]
    w.execute_model(ExecuteModelRequest(sgs, num_lookahead_slots=0))
    rows = {0: [11, 12, -1, -1], 1: [21, -1, -1, -1], 2: [31, 32, 33, -1], 3: [41, 42, 43, 44]}
    w.engine.out_script = lambda step, slots: rows
    order = [decode(sgs[2]), decode(sgs[0]), decode(sgs[1])]          # "d" sits this step out
    req = ExecuteModelRequest(order, num_lookahead_slots=k)
    outs = w.execute_model(req)
    assert w.engine.calls[-1] == ("step", (2, 0, 1))
    assert req.w4a4 is False and w.proposer_calls == k                # toggled around the proposer only (:797-812)
    assert [[g.samples[0].parent_seq_id for g in o.outputs] for o in outs] == [[2, 0, 1]] * 3   # request order; step 4
    assert [o.token_ids() for o in outs] == [[31, 11, 21], [32, 12, -1], [33, -1, -1]]          # is all -1: dropped
    assert w.engine._len_ub[3] == 4 + 3 + 1                           # "d" did not advance
    assert w.engine._len_ub[2] == 4 + 2 + 1 + 3                       # "c": three tokens emitted
    # all four, bonus token for "d": four steps come back
    outs = w.execute_model(ExecuteModelRequest([decode(s) for s in sgs], num_lookahead_slots=k))
    assert len(outs) == 4 and outs[3].token_ids() == [-1, -1, -1, 44]


def test_bonus_token_tracking_and_finished_requests():
    """:1178-1210 -- the set of sequences that got a bonus token in their last step, cleared when a request finishes;
    a finished request frees its slot, which the next prompt takes."""
    k = 2
    w = make_worker(k=k, B=2)
    sgs = [prompt("a", 100, 4), prompt("b", 200, 4)]
    w.execute_model(ExecuteModelRequest(sgs, num_lookahead_slots=0))
    w.engine.out_script = lambda step, slots: {0: [5, 6, 7], 1: [8, -1, -1]}
    w.execute_model(ExecuteModelRequest([decode(s) for s in sgs], num_lookahead_slots=k))
    assert w._seq_with_bonus_token_in_last_step == {100}
    assert dict(w._request_id_seq_id_mapping) == {"a": {100}, "b": {200}}
    w.engine.out_script = lambda step, slots: {0: [5, -1, -1], 1: [8, 9, 10]}
    w.execute_model(ExecuteModelRequest([decode(s) for s in sgs], num_lookahead_slots=k))
    assert w._seq_with_bonus_token_in_last_step == {200}
    with pytest.raises(RuntimeError, match="no free sequence slot"):
        w.execute_model(ExecuteModelRequest([prompt("c", 300, 3)], num_lookahead_slots=0))
    # "b" finishes: its slot (1) is released and "c" moves in
    out = w.execute_model(ExecuteModelRequest([prompt("c", 300, 3)], num_lookahead_slots=0, finished_requests_ids=["b"]))
    assert ("free", 1) in w.engine.calls and w._slots == {"a": 0, "c": 1}
    assert w._seq_with_bonus_token_in_last_step == set() and "b" not in w._request_id_seq_id_mapping
    assert out[0].token_ids() == [1001]
    with pytest.raises(KeyError):
        w.execute_model(ExecuteModelRequest([decode(sgs[1])], num_lookahead_slots=k))     # "b" is gone


def test_speculation_off_paths():
    """k == 0 from the scheduler, all requests with num_speculative_tokens == 0, and speculative_disable_by_batch_size:
    the scorer alone, one token per sequence (:497-538, :666-720)."""
    w = make_worker(k=3, disable_by_batch_size=3)
    sgs = [prompt("a", 1, 4), prompt("b", 2, 6)]
    w.execute_model(ExecuteModelRequest(sgs, num_lookahead_slots=0))
    dec = [decode(s) for s in sgs]
    out = w.execute_model(ExecuteModelRequest(dec, num_lookahead_slots=0))
    assert len(out) == 1 and out[0].token_ids() == [500, 501] and w.engine.calls[-1][0] == "step_no_spec"
    out = w.execute_model(ExecuteModelRequest(dec, num_lookahead_slots=3, running_queue_size=3))
    assert len(out) == 1 and w.engine.calls[-1] == ("step_no_spec", (0, 1)) and w.proposer_calls == 0
    for s in dec:
        s.num_speculative_tokens = 0
    out = w.execute_model(ExecuteModelRequest(dec, num_lookahead_slots=3))
    assert len(out) == 1 and w.engine.calls[-1][0] == "step_no_spec"
    for s in dec:
        s.num_speculative_tokens = None
    w.execute_model(ExecuteModelRequest(dec, num_lookahead_slots=3))
    assert w.engine.calls[-1] == ("step", (0, 1)) and w.proposer_calls == 3


def test_mixed_prompt_and_decode_batch_without_lookahead():
    w = make_worker()
    a = prompt("a", 1, 4)
    w.execute_model(ExecuteModelRequest([a], num_lookahead_slots=0))
    out = w.execute_model(ExecuteModelRequest([prompt("b", 2, 5), decode(a)], num_lookahead_slots=0))
    assert [g.samples[0].parent_seq_id for g in out[0].outputs] == [2, 1] and out[0].token_ids() == [1001, 500]
    assert w.engine.calls[-2:] == [("add", 1, 5, None), ("step_no_spec", (0,))]
    assert w.engine._len_ub == [6, 6, 0, 0]


def test_empty_batch_and_stop_signal():
    w = make_worker()
    assert w.execute_model(ExecuteModelRequest([], num_lookahead_slots=0)) == []
    assert w.execute_model(ExecuteModelRequest([], num_lookahead_slots=3)) == []
    assert w.execute_model(None) == []
    with pytest.raises(NotImplementedError):
        w.get_cache_block_size_bytes()
    with pytest.raises(NotImplementedError):
        create_spec_worker(model_config=CFG, pipeline_parallel_size=2)


def test_scheduler_length_cross_check():
    w = make_worker(k=2)
    a = prompt("a", 1, 4)
    w.execute_model(ExecuteModelRequest([a], num_lookahead_slots=0))
    a.seq_data[1].output_token_ids = [1000]                            # the scheduler appended the first token: 5 == 5
    w.engine.out_script = lambda step, slots: {0: [5, 6, -1]}
    w.execute_model(ExecuteModelRequest([decode(a)], num_lookahead_slots=2))
    a.seq_data[1].output_token_ids = [1000, 5]                          # ... but lost one of the two new tokens
    with pytest.raises(ValueError, match="scheduler holds"):
        w.execute_model(ExecuteModelRequest([decode(a)], num_lookahead_slots=2))


def test_tensor_parallel_control_plane_two_ranks():
    """Driver rank 0 broadcasts the control record + request; rank 1 sits in start_worker_execution_loop() and
    mirrors every call until execute_model(None) (:524-538, :722-755).  Ranks are threads over ThreadComm."""
    from qspec_amd.parallel import TensorParallel, ThreadComm

    class M:   # a model stand-in that only carries the TP context
        pass
    shared = ThreadComm.Shared(2)
    workers = []
    for r in range(2):
        m = M()
        m.tp = TensorParallel(r, 2, None, comm=ThreadComm(shared, r))
        workers.append(make_worker(k=2, model=m, rank=r))
    t = threading.Thread(target=workers[1].start_worker_execution_loop)
    t.start()
    d = workers[0]
    sgs = [prompt("a", 1, 4), prompt("b", 2, 5)]
    d.execute_model(ExecuteModelRequest(sgs, num_lookahead_slots=0))
    d.execute_model(ExecuteModelRequest([decode(s) for s in sgs], num_lookahead_slots=2))
    d.execute_model(ExecuteModelRequest([decode(sgs[1])], num_lookahead_slots=0, finished_requests_ids=["a"]))
    d.execute_model(None)
    t.join(timeout=30)
    assert not t.is_alive()
    assert workers[1].engine.calls == d.engine.calls
    assert d.engine.calls == [("add", 0, 4, None), ("add", 1, 5, None), ("step", (0, 1)), ("free", 0), ("step_no_spec", (1,))]
    assert workers[1]._slots == d._slots == {"b": 1}


# ------------------------------------------------------------------ the reference's binding: vllm_config, output records

def vllm_config(k=3, max_num_seqs=4, tp=1, pp=1, load_format="dummy", model="/nonexistent/Llama3_8B_Instruct_QSpec",
                disable_logprobs=True, disable_by_batch_size=None):
    """What vLLM's WorkerWrapper hands to create_spec_worker (spec_decode_worker.py:53-113), duck-typed."""
    hf = SimpleNamespace(model_type="llama_quarot", hidden_size=1024, intermediate_size=3584, num_attention_heads=8,
                         num_key_value_heads=2, num_hidden_layers=2, vocab_size=2048, rms_norm_eps=1e-5,
                         rope_theta=10000.0, max_position_embeddings=512)
    return SimpleNamespace(
        model_config=SimpleNamespace(hf_config=hf, max_model_len=256, model=model, seed=0, max_logprobs=5),
        cache_config=SimpleNamespace(block_size=16, gpu_memory_utilization=0.9, swap_space_bytes=4 * GiB),
        scheduler_config=SimpleNamespace(max_num_seqs=max_num_seqs, max_num_batched_tokens=2048),
        parallel_config=SimpleNamespace(tensor_parallel_size=tp, pipeline_parallel_size=pp),
        load_config=SimpleNamespace(load_format=load_format),
        speculative_config=SimpleNamespace(num_speculative_tokens=k, speculative_disable_by_batch_size=disable_by_batch_size,
                                           disable_log_stats=True, disable_logprobs=disable_logprobs,
                                           draft_token_acceptance_method="rejection_sampler",
                                           speculative_disable_mqa_scorer=False))


def worker_from_vllm_config(**kw):
    w = create_spec_worker(vllm_config=vllm_config(**kw), local_rank=0, rank=0,
                           distributed_init_method="tcp://127.0.0.1:29555", is_driver_worker=True,
                           device="cpu", model=object(), engine_factory=FakeEngine)
    w._memory_probe = fake_memory_probe
    w.init_device()
    nb, ncpu = w.determine_num_available_blocks()
    w.initialize_cache(nb, ncpu)
    return w, nb, ncpu


def test_factory_binds_vllm_config():
    """create_spec_worker(*args, **kwargs) reads kwargs["vllm_config"] + local_rank / rank / distributed_init_method /
    is_driver_worker as the reference's does (:53-113) -- not a bespoke keyword set that would silently build a synthetic
    8B worker with four slots."""
    w, nb, ncpu = worker_from_vllm_config(k=2, max_num_seqs=6, disable_by_batch_size=5)
    c = w.model_config
    assert (c.hidden_size, c.intermediate_size, c.num_attention_heads, c.num_key_value_heads, c.num_hidden_layers,
            c.vocab_size, c.max_position_embeddings) == (1024, 3584, 8, 2, 2, 2048, 512)
    assert (w.max_num_seqs, w.max_model_len, w.block_size, w.engine.k, w.disable_by_batch_size) == (6, 256, 16, 2, 5)
    assert w.engine.num_blocks == nb and w.rank == 0 and w.max_logprobs == 5
    with pytest.raises(NotImplementedError, match="pipeline parallelism"):
        create_spec_worker(vllm_config=vllm_config(pp=2), local_rank=0, rank=0)
    cfg = vllm_config()
    cfg.speculative_config = None
    with pytest.raises(AssertionError):
        create_spec_worker(vllm_config=cfg, local_rank=0, rank=0)
    cfg = vllm_config()
    cfg.model_config.hf_config.model_type = "qwen2_quarot"
    with pytest.raises(NotImplementedError, match="llama_quarot"):
        create_spec_worker(vllm_config=cfg, local_rank=0, rank=0)


def test_model_path_that_is_no_checkpoint_raises_instead_of_inventing_weights():
    w = create_spec_worker(vllm_config=vllm_config(load_format="auto"), local_rank=0, rank=0, device="cpu",
                           engine_factory=FakeEngine)
    with pytest.raises(FileNotFoundError, match="not a local QSpec checkpoint"):
        w.init_device()


def test_num_available_blocks_from_profiled_memory():
    """:400-426 over vllm/worker/worker.py:176-265: (total x utilisation - used - peak - 2 x peak) / block bytes, the
    scorer's count un-split (:421-423); initialize_cache builds ONE cache of exactly that many blocks and refuses a
    count that cannot hold max_model_len (raise_if_cache_size_invalid, worker.py:541-558)."""
    w, nb, ncpu = worker_from_vllm_config()
    block = 2 * 2 * 16 * 2 * 128 * 2                         # K+V x layers x block x kv heads x head_dim x fp16
    assert w.cache_block_size_bytes() == block
    available = 288 * GiB * 0.9 - (88 * GiB + 2 * GiB) - 2 * 2 * GiB
    assert nb == int(available // block) and ncpu == 4 * GiB // block
    assert w.memory_profile["torch_peak_increase"] == 2 * GiB and w.memory_profile["avoid_oom_memory"] == 4 * GiB
    assert w.engine.num_blocks == nb
    with pytest.raises(ValueError, match="No available memory"):
        w.initialize_cache(0, 0)
    with pytest.raises(ValueError, match="max seq len"):
        w.initialize_cache(15, 0)                            # 15 x 16 = 240 tokens < max_model_len 256
    w._memory_probe = lambda: (1 * GiB, 288 * GiB, 2 * GiB)  # a card that is already full
    assert w.determine_num_available_blocks()[0] == 0


class Sequence:
    """The slice of vllm.sequence.Sequence the multi-step output processor touches."""

    def __init__(self, seq_id, prompt_len):
        self.seq_id, self.prompt_len, self.output, self.logprobs = seq_id, prompt_len, [], []

    def get_output_len(self):
        return len(self.output)


def process_outputs(seq, outputs, sampling_params, eos_token_id):
    """Restatement of MultiStepOutputProcessor.process_outputs / _process_seq_outputs
    (vllm/engine/output_processor/multi_step.py:100-176): takes `output.samples[0]` of every step, asserts the parent
    sequence id, drops invalid (-1) tokens, truncates to max_tokens and after EOS, appends."""
    assert all(isinstance(o, CompletionSequenceGroupOutput) for o in outputs)
    assert all(seq.seq_id == o.samples[0].parent_seq_id for o in outputs)
    samples = [o.samples[0] for o in outputs]
    valid = [s for s in samples if s.output_token != -1]
    if not valid:
        return
    ids = [s.output_token for s in valid]
    lps = [s.logprobs for s in valid]
    remaining = sampling_params.max_tokens - (seq.get_output_len() + len(ids))
    if remaining < 0:
        ids = ids[:remaining]
    if not sampling_params.ignore_eos:
        for i, t in enumerate(ids):
            if t == eos_token_id:
                ids = ids[:i + 1]
                break
    for t, lp in zip(ids, lps):
        assert t in lp and isinstance(lp[t], Logprob)            # vLLM logprobs always include the sampled token
        seq.output.append(t)
        seq.logprobs.append(lp)


def test_outputs_feed_the_multi_step_output_processor():
    """The worker's List[SamplerOutput] is what LLMEngine hands to MultiStepOutputProcessor per sequence group:
    outputs[i] of every step (spec_decode_worker.py:1023-1046 -> multi_step.py:100-176)."""
    EOS = 99
    w, _, _ = worker_from_vllm_config(k=3)
    sp = SamplingParams(max_tokens=5)
    sgs = [SequenceGroupMetadata(r, True, {sid: SequenceData(list(range(n)))}, sampling_params=sp)
           for r, sid, n in (("a", 7, 4), ("b", 8, 6), ("c", 9, 5))]
    seqs = {sg.request_id: Sequence(next(iter(sg.seq_data)), 0) for sg in sgs}
    out = w.execute_model(ExecuteModelRequest(sgs, num_lookahead_slots=0))
    assert len(out) == 1 and isinstance(out[0].outputs[0], CompletionSequenceGroupOutput)
    first = out[0].outputs[0].samples[0]
    assert isinstance(first, SequenceOutput) and first.logprobs == {1000: Logprob(0.0, -1)}     # :652-660
    assert out[0].sampled_token_ids is None and out[0].sampled_token_probs is None and out[0].logprobs is None
    for i, sg in enumerate(sgs):
        process_outputs(seqs[sg.request_id], [o.outputs[i] for o in out], sp, EOS)
    assert [seqs[r].output for r in "abc"] == [[1000], [1001], [1002]]
    # one speculative step: "a" gets all k + bonus, "b" hits EOS at its second token, "c" emits one token
    w.engine.out_script = lambda step, slots: {0: [11, 12, 13, 14], 1: [21, EOS, 23, -1], 2: [31, -1, -1, -1]}
    dec = [SequenceGroupMetadata(s.request_id, False, s.seq_data, sampling_params=sp) for s in sgs]
    outs = w.execute_model(ExecuteModelRequest(dec, num_lookahead_slots=3))
    assert len(outs) == 4 and all(len(o.outputs) == 3 for o in outs)
    s0 = outs[0].outputs[0].samples[0]
    assert (s0.parent_seq_id, s0.output_token, s0.logprobs) == (7, 11, {11: Logprob(0.0, -1)})   # dummy logprobs :1083
    for i, sg in enumerate(dec):
        process_outputs(seqs[sg.request_id], [o.outputs[i] for o in outs], sp, EOS)
    assert seqs["a"].output == [1000, 11, 12, 13, 14]            # max_tokens = 5 reached exactly
    assert seqs["b"].output == [1001, 21, EOS]                   # truncated after EOS
    assert seqs["c"].output == [1002, 31]
    # a step in which nothing follows position 0: the list stops at the first all-(-1) position (:1023-1026)
    w.engine.out_script = lambda step, slots: {2: [32, -1, -1, -1]}
    outs = w.execute_model(ExecuteModelRequest([dec[2]], num_lookahead_slots=3, finished_requests_ids=["a", "b"]))
    assert len(outs) == 1 and outs[0].token_ids() == [32]
    # max_tokens truncation inside a step
    seqs["c"].output = [1, 2, 3]
    w.engine.out_script = lambda step, slots: {2: [41, 42, 43, 44]}
    outs = w.execute_model(ExecuteModelRequest([dec[2]], num_lookahead_slots=3))
    process_outputs(seqs["c"], [o.outputs[0] for o in outs], sp, EOS)
    assert seqs["c"].output == [1, 2, 3, 41, 42]


def test_requested_logprobs_with_dummy_lists_and_empty_prompt_chunk():
    """num_logprobs > 0 while logprobs are disabled during speculation: the top-k entries of the dummy lists are None and
    are dropped (util.py:78-86); a sequence group with do_sample == False gets an empty output (:625-631); prompt
    logprobs requested -> one dummy dict per prompt token but the first (:637-650)."""
    w, _, _ = worker_from_vllm_config(k=2)
    sp = SamplingParams(logprobs=3, prompt_logprobs=1)
    a = SequenceGroupMetadata("a", True, {1: SequenceData([5, 6, 7])}, sampling_params=sp)
    b = SequenceGroupMetadata("b", True, {2: SequenceData([5, 6, 7, 8])}, sampling_params=SamplingParams(), do_sample=False)
    out = w.execute_model(ExecuteModelRequest([a, b], num_lookahead_slots=0))
    assert out[0].outputs[1].samples == [] and out[0].outputs[1].prompt_logprobs is None
    assert out[0].outputs[0].prompt_logprobs == [{6: Logprob(0.0, -1)}, {7: Logprob(0.0, -1)}]
    w.engine.out_script = lambda step, slots: {0: [9, 10, -1]}
    outs = w.execute_model(ExecuteModelRequest([SequenceGroupMetadata("a", False, a.seq_data, sampling_params=sp)],
                                               num_lookahead_slots=2))
    assert [o.outputs[0].samples[0].logprobs for o in outs] == [{9: Logprob(0.0, -1)}, {10: Logprob(0.0, -1)}]


def test_request_without_speculation_inside_a_speculative_step():
    """A request with num_speculative_tokens == 0 in a speculative batch gets the target's single token (proposal length
    0, top1_proposer.py:103-135): it sits out the cycle and takes the scorer-only step; the others speculate."""
    w = make_worker(k=2)
    sgs = [prompt("a", 1, 4), prompt("b", 2, 6)]
    w.execute_model(ExecuteModelRequest(sgs, num_lookahead_slots=0))
    dec = [decode(s) for s in sgs]
    dec[1].num_speculative_tokens = 0
    w.engine.out_script = lambda step, slots: {0: [5, 6, 7]} if step == 0 else None
    outs = w.execute_model(ExecuteModelRequest(dec, num_lookahead_slots=2))
    assert w.engine.calls[-2:] == [("step", (0,)), ("step_no_spec", (1,))]
    assert [o.token_ids() for o in outs] == [[5, 501], [6, -1], [7, -1]]
    with pytest.raises(ValueError, match="captured for"):
        w.execute_model(ExecuteModelRequest([decode(s) for s in sgs], num_lookahead_slots=4))


def test_device_handoff_timeout_replays_once_then_raises():
    """A non-zero error word read with the cycle's output: the cycle is re-run once without device-side hand-offs
    (engine.recover); only a second failure raises, and as a distinct exception type."""
    w = make_worker(k=2)
    a = prompt("a", 1, 4)
    w.execute_model(ExecuteModelRequest([a], num_lookahead_slots=0))
    w.engine.out_script = lambda step, slots: {0: [5, 6, -1]}
    w.engine.err_script = [1]                                # the first read reports a timed-out hand-off
    outs = w.execute_model(ExecuteModelRequest([decode(a)], num_lookahead_slots=2))
    assert ("recover",) in w.engine.calls and [o.token_ids() for o in outs] == [[5], [6]]
    w.engine.err_script = [1, 1]
    a.seq_data[1].output_token_ids = []
    with pytest.raises(DeviceHandoffTimeout):
        w.execute_model(ExecuteModelRequest([decode(a)], num_lookahead_slots=2))


def test_out_of_cycle_error_words_fail_prompt_passes_and_no_spec_steps():
    """Prompt passes and non-speculative steps run outside the captured cycle: the worker checks every hand-off workspace's
    sticky word from the host behind them (they end in a host read anyway) and fails the call -- no silent invalid tokens."""
    w = make_worker(k=2)
    a = prompt("a", 1, 4)
    w.execute_model(ExecuteModelRequest([a], num_lookahead_slots=0))
    w.engine.error_flag = lambda: 1
    with pytest.raises(DeviceHandoffTimeout, match="non-speculative"):
        w.execute_model(ExecuteModelRequest([decode(a)], num_lookahead_slots=0))
    with pytest.raises(DeviceHandoffTimeout, match="prompt pass"):
        w.execute_model(ExecuteModelRequest([prompt("b", 2, 5)], num_lookahead_slots=0))


def test_engine_free_slot_forgets_a_brought_block_table():
    """with fewer KV blocks than max_num_seqs x blocks_per_seq every request brings its table; a finished
    request's table (and the capacity it gave the slot) must not survive free_slot(), or a later admission WITHOUT a table
    would pass validation against it and write KV into blocks the scheduler has handed to someone else.  Host-side
    bookkeeping only: the engine is built on CPU tensors, nothing is launched (validation precedes the first kernel)."""
    from types import SimpleNamespace
    from qspec_amd.spec_decode import QSpecEngine
    cfg = QuarotLlamaConfig(256, 512, 2, 2, 1, 64, 1e-5, 10000.0, 512, "cpu-shell")
    model = SimpleNamespace(config=cfg, device=torch.device("cpu"), tp=None)
    eng = QSpecEngine(model, 2, 2, max_model_len=64, block_size=16, max_new_tokens=16, use_graph=False, num_blocks=3)
    assert eng._capacity == [0, 0]                                   # brought-table mode
    eng.set_block_table(0, [2, 1])
    assert eng._capacity[0] == 32 and eng.block_tables[0, :2].tolist() == [2, 1]
    eng._len_ub[0] = 9                                               # (as an admitted request)
    eng.free_slot(0)
    assert eng._capacity[0] == 0 and eng._bt_host[0] is None and eng.block_tables[0].tolist() == [0, 0, 0, 0]
    with pytest.raises(ValueError, match="no block table|capacity 0"):
        eng.add_sequence(0, [1, 2, 3])
    with pytest.raises(ValueError, match="no block table"):
        eng.add_sequences_to([0, 1], [[1, 2, 3], [4, 5]])
    # contiguous mode: a slot that was given a table goes back to its own default range
    eng = QSpecEngine(model, 2, 2, max_model_len=64, block_size=16, max_new_tokens=16, use_graph=False)
    eng.set_block_table(1, [0, 1])
    eng._len_ub[1] = 5
    eng.free_slot(1)
    assert eng.block_tables[1].tolist() == [4, 5, 6, 7] and eng._capacity[1] == 64 and eng._bt_host[1] is None


def test_acceptance_method_selects_the_sampler():
    """spec_decode_worker.py:95-110: draft_token_acceptance_method picks RejectionSampler (default) or
    TypicalAcceptanceSampler with the two posterior parameters; anything else is refused."""
    from qspec_amd.spec_decode import TypicalAcceptanceSampler
    w = make_worker()
    assert w.engine.acceptance_sampler is None                       # the engine's default: RejectionSampler
    spec = SpeculativeConfig(3, draft_token_acceptance_method="typical_acceptance_sampler",
                             typical_acceptance_sampler_posterior_threshold=0.2, typical_acceptance_sampler_posterior_alpha=0.5)
    w = create_spec_worker(model_config=CFG, model=object(), speculative_config=spec, max_num_seqs=2, max_model_len=256,
                           block_size=16, device="cpu", engine_factory=FakeEngine, disable_log_stats=True,
                           memory_probe=fake_memory_probe)
    w.init_device()
    nb, _ = w.determine_num_available_blocks()
    w.initialize_cache(nb, 0)
    s = w.engine.acceptance_sampler
    assert isinstance(s, TypicalAcceptanceSampler) and (s._posterior_threshold, s._posterior_alpha) == (0.2, 0.5)
    with pytest.raises(ValueError, match="draft_token_acceptance_method"):
        create_spec_worker(model_config=CFG, model=object(), speculative_config=SpeculativeConfig(3, draft_token_acceptance_method="x"),
                           engine_factory=FakeEngine, device="cpu")


def test_sampling_params_reach_the_engine_before_the_prompt_pass():
    """temperature / top_k / top_p of a request (vllm/sampling_params.py) are handed to the engine at admission, before the
    prompt pass (its first token is sampled with them); no SamplingParams = greedy."""
    w = make_worker()
    a = SequenceGroupMetadata("a", True, {1: SequenceData([1, 2, 3])}, sampling_params=SamplingParams(temperature=0.7, top_k=40, top_p=0.95))
    b = SequenceGroupMetadata("b", True, {2: SequenceData([4, 5])})
    w.execute_model(ExecuteModelRequest([a, b], num_lookahead_slots=0))
    assert w.engine.sampling == {0: (0.7, 40, 0.95), 1: (0.0, -1, 1.0)}
    order = [c[0] for c in w.engine.calls]
    assert order == ["add", "add"]


def test_failed_admission_gives_the_slots_back():
    """If the engine refuses an admission, worker and engine must still agree that the slots are free."""
    w = make_worker()

    def failing(slots, prompts, block_tables=None):
        w.engine.add_sequence(slots[0], prompts[0])          # the first request got in ...
        raise ValueError("prompt does not fit")              # ... the second did not
    orig = w.engine.add_sequences_to
    w.engine.add_sequences_to = failing
    with pytest.raises(ValueError, match="does not fit"):
        w.execute_model(ExecuteModelRequest([prompt("a", 1, 4), prompt("b", 2, 9)], num_lookahead_slots=0))
    assert w._slots == {} and w.engine._len_ub == [0, 0, 0, 0]
    w.engine.add_sequences_to = orig
    out = w.execute_model(ExecuteModelRequest([prompt("a", 1, 4)], num_lookahead_slots=0))
    assert out[0].token_ids() == [1000]


@pytest.mark.parametrize("k", [1, 2, 6])
@pytest.mark.parametrize("batch_size", [1, 2, 32])
def test_correctly_formats_output(k, batch_size):
    """tests/spec_decode/test_spec_decode_worker.py:236-370 (`test_correctly_formats_output`): a rejection-sampler output
    [batch, k + 1] with every row's tokens followed by -1 padding comes back as one SamplerOutput per step holding, per
    sequence, that step's token under the sequence's own id -- transposed, complete up to the last step in which any
    sequence emitted a token."""
    gen = torch.Generator().manual_seed(k * 100 + batch_size)
    w = make_worker(k=k, B=batch_size)
    sgs = [prompt(f"r{i}", 1000 + i, 3 + i % 5) for i in range(batch_size)]
    w.execute_model(ExecuteModelRequest(sgs, num_lookahead_slots=0))
    toks = torch.randint(0, 32000, (batch_size, k + 1), generator=gen)
    n_emit = torch.randint(1, k + 2, (batch_size,), generator=gen)      # every sequence emits 1 .. k+1 tokens
    toks[torch.arange(k + 1)[None, :] >= n_emit[:, None]] = -1
    rows = {b: toks[b].tolist() for b in range(batch_size)}
    w.engine.out_script = lambda step, slots: rows
    outs = w.execute_model(ExecuteModelRequest([decode(s) for s in sgs], num_lookahead_slots=k))
    steps = int(n_emit.max())
    assert len(outs) == steps
    for j, o in enumerate(outs):
        assert [g.samples[0].parent_seq_id for g in o.outputs] == [1000 + i for i in range(batch_size)]
        assert o.token_ids() == toks[:, j].tolist()
    # the bonus-token set (:1190-1210): exactly the sequences whose last position holds a token
    assert w._seq_with_bonus_token_in_last_step == {1000 + i for i in range(batch_size) if int(toks[i, k]) != -1}
