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
from qspec_amd.spec_decode.worker import (ExecuteModelRequest, SequenceData, SequenceGroupMetadata, SpeculativeConfig,
                                          create_spec_worker)


class FakeSampler:
    def __init__(self):
        self.counters = torch.zeros(3, dtype=torch.long)
        self.num_accepted_tokens = self.counters[0]
        self.num_emitted_tokens = self.counters[1]
        self.num_draft_tokens = 0


class FakeEngine:
    """The QSpecEngine surface the worker uses, with scripted outputs: out_script(step_index, slots) -> [B, k+1]."""

    def __init__(self, model, k, B, max_model_len, block_size, seed=0):
        self.k, self.B = k, B
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


CFG = QuarotLlamaConfig(1024, 3584, 8, 2, 2, 2048, 1e-5, 10000.0, 512, "tiny")


def make_worker(k=3, B=4, disable_by_batch_size=None, model=None, rank=0):
    w = create_spec_worker(model_config=CFG, model=model if model is not None else object(),
                           speculative_config=SpeculativeConfig(k, speculative_disable_by_batch_size=disable_by_batch_size),
                           max_num_seqs=B, max_model_len=256, block_size=16, device="cpu", engine_factory=FakeEngine,
                           disable_log_stats=True, rank=rank)
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
    assert len(out) == 1 and out[0].request_ids == ["a", "b"]
    assert out[0].sampled_token_ids.tolist() == [1000, 1001]          # the target's first token of each prompt
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
    assert [o.request_ids for o in outs] == [["c", "a", "b"]] * 3     # step 4 is all -1 for these three: dropped
    assert [o.sampled_token_ids.tolist() for o in outs] == [[31, 11, 21], [32, 12, -1], [33, -1, -1]]
    assert w.engine._len_ub[3] == 4 + 3 + 1                           # "d" did not advance
    assert w.engine._len_ub[2] == 4 + 2 + 1 + 3                       # "c": three tokens emitted
    # all four, bonus token for "d": four steps come back
    outs = w.execute_model(ExecuteModelRequest([decode(s) for s in sgs], num_lookahead_slots=k))
    assert len(outs) == 4 and outs[3].sampled_token_ids.tolist() == [-1, -1, -1, 44]


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
    assert out[0].sampled_token_ids.tolist() == [1001]
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
    assert len(out) == 1 and out[0].sampled_token_ids.tolist() == [500, 501] and w.engine.calls[-1][0] == "step_no_spec"
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
    assert out[0].request_ids == ["b", "a"] and out[0].sampled_token_ids.tolist() == [1001, 500]
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
