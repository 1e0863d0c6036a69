"""World-size-2 gloo tests of the tensor-parallel plumbing (CPU only): shard ranges, all-reduce of zero-padded
column shards == concatenation, vocab all-gather with uneven ranges."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from qspec_amd.parallel import TensorParallel, shard_range


def test_shard_range_covers_and_aligns():
    for n, world, align in ((4096, 8, 128), (14336, 8, 128), (14336, 8, 32), (128256, 8, 16), (2048, 3, 16), (3584, 2, 32)):
        prev = 0
        for r in range(world):
            lo, hi = shard_range(n, world, r, align)
            assert lo == prev and lo % align == 0 and hi % align == 0 and hi >= lo
            prev = hi
        assert prev == n


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tp = TensorParallel(rank, world, None)
    T, I, V = 3, 64, 16 * 5   # V: 5 tiles over 2 ranks -> uneven vocab ranges
    full = torch.arange(T * I, dtype=torch.float32).view(T, I).to(torch.float16)
    c0, c1 = tp.channel_range(I)
    act = torch.zeros(T, I, dtype=torch.float16)
    act[:, c0:c1] = full[:, c0:c1]
    tp.all_reduce(act)
    ok1 = torch.equal(act, full)
    # row-parallel partial sums
    x = torch.full((T, 8), float(rank + 1))
    tp.all_reduce(x)
    ok2 = bool((x == sum(range(1, world + 1))).all())
    logits_full = torch.arange(T * V, dtype=torch.float32).view(T, V).to(torch.float16)
    v0, v1 = tp.vocab_range(V)
    out = torch.empty(T, V, dtype=torch.float16)
    tp.all_gather_vocab(logits_full[:, v0:v1].contiguous(), out, V)
    ok3 = torch.equal(out, logits_full)
    # the draft pass's plan: the logits all-gather is timed on THIS job's communicator (MAX over the ranks: every rank
    # must see the same number, or the ranks would take different plans and issue different collectives)
    from qspec_amd.parallel import agree_all, measure_vocab_gather_us
    us = measure_vocab_gather_us(tp, 4, 16 * 64, "cpu", iters=3)
    same = [None] * world
    dist.all_gather_object(same, us)
    ok4 = us > 0 and all(v == same[0] for v in same) and agree_all(tp, True, "cpu") and not agree_all(tp, rank == 0, "cpu")
    ret[rank] = (ok1 and ok4, ok2, ok3, (v0, v1))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gloo_world2_collectives():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        ok1, ok2, ok3, _ = ret[r]
        assert ok1 and ok2 and ok3, (r, ret[r])
    assert ret[0][3] != ret[1][3] and ret[0][3][1] == ret[1][3][0]


def test_tp_plan_shards_layers_only_when_it_pays(monkeypatch):
    """Llama-3-8B layers (109 MB) stay replicated at every world size; Llama-3-70B layers (436 MB) shard at 8."""
    from qspec_amd.parallel import shard_layers_pays
    monkeypatch.delenv("QSPEC_TP_LAYERS", raising=False)
    l8b = (6144 + 4096 + 28672) * 2048 + 4096 * 7168
    l70b = (10240 + 8192 + 57344) * 4096 + 8192 * 14336
    assert not any(shard_layers_pays(l8b, w) for w in (1, 2, 4, 8))
    assert shard_layers_pays(l70b, 8) and not shard_layers_pays(l70b, 1)
    monkeypatch.setenv("QSPEC_TP_LAYERS", "1")
    assert shard_layers_pays(l8b, 2)
    monkeypatch.setenv("QSPEC_TP_LAYERS", "0")
    assert not shard_layers_pays(l70b, 8)


def test_tp_plan_draft_vocab_parallel_only_when_it_pays():
    """The draft pass's lm_head goes vocab-parallel iff the stream it saves exceeds the measured logits all-gather:
    Llama-3-8B's 1.05 GB head at 8 ranks saves ~143 us per forward -> pays against a 40 us gather, not against 200 us;
    a 2048 x 1024 test head never pays."""
    from qspec_amd.parallel import shard_draft_vocab_pays
    head = 128256 * 4096 * 2
    assert shard_draft_vocab_pays(head, 8, 40.0) and not shard_draft_vocab_pays(head, 8, 200.0)
    assert shard_draft_vocab_pays(head, 2, 40.0) and not shard_draft_vocab_pays(head, 2, 100.0)
    assert not shard_draft_vocab_pays(2048 * 1024 * 2, 8, 5.0)


def test_thread_ranks_collectives_and_column_gathers():
    """ThreadComm (ranks as threads of one process; what the 8-rank full-width TP test uses on the GPU): rank-order fp32
    all-reduce, all-gather of even and uneven column ranges, object broadcast -- on CPU tensors."""
    import threading
    from qspec_amd.parallel import ThreadComm
    world = 4
    shared = ThreadComm.Shared(world)
    out, errs = {}, []

    def body(r):
        try:
            tp = TensorParallel(r, world, None, comm=ThreadComm(shared, r))
            x = torch.full((3, 8), float(r + 1))
            tp.all_reduce(x)
            I = 32 * 6                                   # 6 units of 32 over 4 ranks: 2, 2, 1, 1 -> uneven ranges
            full = torch.arange(3 * I, dtype=torch.float32).view(3, I)
            c0, c1 = tp.channel_range(I)
            act = torch.zeros(3, I)
            act[:, c0:c1] = full[:, c0:c1]
            tp.all_gather_channels(act, I)
            V = 16 * 8                                   # even vocabulary ranges
            logits = torch.arange(2 * V, dtype=torch.float32).view(2, V)
            v0, v1 = tp.vocab_range(V)
            got = torch.empty(2, V)
            tp.all_gather_vocab(logits[:, v0:v1].contiguous(), got, V)
            obj = tp.broadcast_object({"k": 3} if r == 0 else None, src=0)
            out[r] = (bool((x == 10.0).all()), torch.equal(act, full), torch.equal(got, logits), obj)
        except BaseException as exc:  # noqa: BLE001
            errs.append((r, repr(exc)))
            shared.barrier.abort()
    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(timeout=60) for t in ts]
    assert not errs, errs
    assert all(out[r] == (True, True, True, {"k": 3}) for r in range(world)), out
