"""Tensor parallelism for the verify pass (option A).

The reference QSpec model has no TP (plain nn.Embedding / Linear4bit / HF lm_head).  Design here:

  * every rank holds the FULL packed int4 buffer (Llama-3-70B: 35 GB << 288 GB), the full KV cache and the
    full embedding / lm_head; sharding is a *view* (K range or channel range) of the same bytes;
  * the DRAFT pass runs replicated, with zero collectives: all kernels are deterministic, every rank computes
    bit-identical tokens (the Philox state of the rejection sampler is seeded identically);
  * the VERIFY pass (decode-sized M) shards the three weight-heavy GEMMs that tolerate it:
        o_proj    row-parallel  (K range of the Hadamard output)          -> all-reduce [T, H]  fp16
        gate_up   column-parallel (channel range, silu*up fused)          -> all-gather [T, I/tp] fp16 (exact)
        down_proj row-parallel  (K range of the Hadamard output)          -> all-reduce [T, H]  fp16
        lm_head   vocab-parallel                                          -> all-gather [T, V/tp]
    qkv_proj + RoPE + KV write + attention stay replicated (the online Hadamards mix all heads / all of I and
    the per-token abs-max needs the whole row, so their inputs must be complete on every rank anyway; replicated
    attention also keeps every rank's KV cache identical without a KV exchange).
  * collectives: RCCL through torch.distributed (backend "nccl" on ROCm), enqueued on the current stream and
    captured into the cycle's hipGraph.  Messages are tiny (128 KB .. 460 KB): latency-bound over xGMI.

`gloo` is supported for tests (CPU tensors, or GPU tensors staged through the host).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, world: int, rank: int, align: int) -> Tuple[int, int]:
    """[lo, hi) of rank's contiguous shard of n items, every boundary a multiple of `align`."""
    assert n % align == 0
    units = n // align
    base, rem = divmod(units, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo * align, hi * align


# Planning constants for shard_layers_pays(): what a CU-saturating weight stream reaches on MI355X (measured)
# and a conservative cost of one small RCCL collective inside the captured graph (128-460 KB over xGMI: latency-bound;
# NOT measured here -- no multi-GPU box in this round; QSPEC_TP_LAYERS=0/1 overrides the plan).
STREAM_BYTES_PER_US = 4.5e6
LM_HEAD_BYTES_PER_US = 6.4e6     # the lm_head's fp16 stream through LDS-DMA: 1.05 GB in 162 us
COLLECTIVE_US = {2: 12.0, 4: 18.0, 8: 25.0}


def shard_layers_pays(layer_weight_bytes: int, world: int) -> bool:
    """Sharding the verify pass's o_proj / gate_up / down_proj costs three collectives per layer and saves
    (1 - 1/world) of the layer's weight stream.  At decode-sized M that only pays when the layer is big:
    Llama-3-8B: 109 MB -> 12-21 us saved against 36-75 us of collectives (replicated layers, vocab-parallel lm_head
    only); Llama-3-70B at 8 GPUs: 436 MB -> 85 us saved against ~75 us."""
    import os
    forced = os.environ.get("QSPEC_TP_LAYERS")
    if forced is not None:
        return forced != "0"
    if world < 2:
        return False
    t_coll = COLLECTIVE_US.get(world, 12.0 + 2.0 * world)
    saved = layer_weight_bytes * (1.0 - 1.0 / world) / STREAM_BYTES_PER_US
    return saved > 3.0 * t_coll


class TorchDistComm:
    """Collectives through torch.distributed: RCCL (backend "nccl" on ROCm) on device tensors, enqueued on the current
    stream; `gloo` for tests (device tensors are staged through the host)."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.backend = dist.get_backend(group)

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        if self.backend == "gloo" and t.is_cuda:
            c = t.cpu()
            dist.all_reduce(c, group=self.group)
            t.copy_(c)
        else:
            dist.all_reduce(t, group=self.group)
        return t

    def all_gather(self, recv: torch.Tensor, send: torch.Tensor) -> torch.Tensor:
        """recv [world * n] <- every rank's send [n], rank-major."""
        if self.backend == "gloo" and send.is_cuda:
            rc = recv.cpu()
            dist.all_gather_into_tensor(rc, send.cpu(), group=self.group)
            recv.copy_(rc)
        else:
            dist.all_gather_into_tensor(recv, send, group=self.group)
        return recv

    def broadcast_object(self, obj, src: int = 0):
        box = [obj]
        dist.broadcast_object_list(box, src=src, group=self.group)
        return box[0]


class OneShotComm(TorchDistComm):
    """TorchDistComm whose small fp32 all-reduces (the [T, H] row-parallel partials of the verify pass) go through the
    one-shot push all-reduce over IPC-mapped peer buffers (csrc/comm.hip; the contract of vllm's custom all-reduce,
    custom_all_reduce.py:50-56,242-255); everything else, and anything above `max_bytes`, stays on the library
    collective.  Opt-in (QSPEC_ONESHOT_AR=1, or constructed directly): NOT yet measured on a multi-GPU box -- it is
    exercised with two processes on one GPU (tests/oneshot_check.py)."""

    def __init__(self, rank: int, world: int, group: Optional[dist.ProcessGroup] = None, max_bytes: int = 2 << 20):
        super().__init__(group)
        import ctypes
        from . import _lib
        self._lib = _lib.load()
        self.rank, self.world, self.max_bytes = rank, world, max_bytes
        ctx = ctypes.c_void_p()
        rc = self._lib.qspec_oneshot_create(rank, world, max_bytes, ctypes.byref(ctx))
        if rc != 0:
            raise RuntimeError(f"qspec_oneshot_create failed ({rc})")
        self._ctx = ctx
        hb = self._lib.qspec_oneshot_handle_bytes()
        mine = ctypes.create_string_buffer(hb)
        if self._lib.qspec_oneshot_local_handle(ctx, mine) != 0:
            raise RuntimeError("hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set on this pool)")
        handles = [None] * world
        dist.all_gather_object(handles, mine.raw, group=group)
        rc = self._lib.qspec_oneshot_open_peers(ctx, b"".join(handles))
        if rc != 0:
            raise RuntimeError(f"hipIpcOpenMemHandle failed for rank {rc - 2}")
        dist.barrier(group=group)
        self.backend = "oneshot+" + self.backend

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        if t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() * 4 <= self.max_bytes \
                and t.data_ptr() % 16 == 0 and t.numel() % 4 == 0:
            rc = self._lib.qspec_oneshot_all_reduce_f32(self._ctx, t.data_ptr(), t.numel(),
                                                        torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                raise RuntimeError(f"qspec_oneshot_all_reduce_f32 failed ({rc})")
            return t
        if self.backend.endswith("gloo") and t.is_cuda:
            c = t.cpu()
            dist.all_reduce(c, group=self.group)
            t.copy_(c)
            return t
        dist.all_reduce(t, group=self.group)
        return t

    def all_gather(self, recv: torch.Tensor, send: torch.Tensor) -> torch.Tensor:
        if self.backend.endswith("gloo") and send.is_cuda:
            rc = recv.cpu()
            dist.all_gather_into_tensor(rc, send.cpu(), group=self.group)
            recv.copy_(rc)
            return recv
        dist.all_gather_into_tensor(recv, send, group=self.group)
        return recv

    def error(self) -> int:
        """Sticky error word read from the host (synchronises the device): out-of-band check."""
        return int(self._lib.qspec_oneshot_error(self._ctx))

    def error_word_address(self) -> int:
        """Device address of the sticky error word: the engine collects it with the cycle's output (no host sync) and
        all-reduces the collected word, so that every rank takes the same decision."""
        return int(self._lib.qspec_oneshot_error_word(self._ctx) or 0)

    def close(self):
        if self._ctx is not None:
            self._lib.qspec_oneshot_destroy(self._ctx)
            self._ctx = None


class ThreadComm:
    """`world` ranks as threads of ONE process sharing one device (tests: a GPU box admits at most 6 processes on
    its card, so the 8-rank shard ranges of Llama-3-70B are exercised in-process).  Every rank thread runs on a stream
    of its own (so that the per-(device, stream) workspaces of qspec_amd.ops are per rank, as with one process per GPU);
    the exchange is host-synchronised: a rank drains its stream before it publishes a tensor or lets the others go on.
    Not capturable, not fast: a test communicator."""

    class Shared:
        def __init__(self, world: int):
            import threading
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world
            self.result = None
            self.bcast = None    # broadcast_object's own field: must not alias an in-flight all_reduce result

    def __init__(self, shared: "ThreadComm.Shared", rank: int):
        self.sh, self.rank, self.backend = shared, rank, "threads"

    def _sync_wait(self):
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()
        self.sh.barrier.wait()

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        sh = self.sh
        sh.slots[self.rank] = t
        self._sync_wait()
        if self.rank == 0:   # fixed rank order, fp32 accumulate, one rounding: what a one-shot all-reduce does
            acc = sh.slots[0].float()
            for r in range(1, sh.world):
                acc = acc + sh.slots[r].float()
            sh.result = acc.to(t.dtype)
        self._sync_wait()
        t.copy_(sh.result)
        self._sync_wait()
        return t

    def all_gather(self, recv: torch.Tensor, send: torch.Tensor) -> torch.Tensor:
        sh = self.sh
        sh.slots[self.rank] = send
        self._sync_wait()
        n = send.numel()
        flat = recv.view(-1)
        for r in range(sh.world):
            flat[r * n:(r + 1) * n].copy_(sh.slots[r].reshape(-1))
        self._sync_wait()
        return recv

    def broadcast_object(self, obj, src: int = 0):
        sh = self.sh
        if self.rank == src:
            sh.bcast = obj
        sh.barrier.wait()
        out = sh.bcast
        sh.barrier.wait()
        return out


class TensorParallel:
    def __init__(self, rank: int, world: int, group: Optional[dist.ProcessGroup] = None, shard_layers: bool = True,
                 comm=None):
        self.rank, self.world, self.group = rank, world, group
        if comm is None and world > 1:
            import os
            if os.environ.get("QSPEC_ONESHOT_AR", "0") == "1":
                comm = OneShotComm(rank, world, group)
            else:
                comm = TorchDistComm(group)
        self.comm = comm
        self.backend = comm.backend if comm is not None else "none"
        # False: the decoder layers of the verify pass stay replicated (no collectives); lm_head stays vocab-parallel
        self.shard_layers = shard_layers
        # Draft pass: the decoder layers always run replicated (no collective touches an activation: the online Hadamards
        # and the per-token abs-max need whole rows), but the lm_head + sampler front end -- 4 x 160 us of the 7.5 ms
        # cycle, the one draft-side piece that shards without touching a Hadamard row -- may go vocab-parallel: every
        # rank streams V / world rows of lm_head, the fp16 logit slices are all-gathered ([B, V] fp16: 1 MB at bs = 4) and
        # every rank runs the same softmax / argmax on the same bits (logits_processor.py:104-107).  Whether that pays
        # is a measured decision (attach_tp); QSPEC_TP_DRAFT_VOCAB=0/1 forces it.
        self.shard_draft_vocab = False
        self._bufs = {}   # exchange buffers, allocated once per shape (nothing is allocated inside a captured cycle)

    def _buf(self, key, shape, dtype, device):
        k = (key, tuple(shape), dtype, str(device))
        if k not in self._bufs:
            self._bufs[k] = torch.empty(*shape, dtype=dtype, device=device)
        return self._bufs[k]

    def k_range(self, K: int) -> Tuple[int, int]:
        """Row-parallel K range: multiples of 128 (one MFMA step of the W4A16 kernel)."""
        return shard_range(K, self.world, self.rank, 128)

    def channel_range(self, I: int) -> Tuple[int, int]:
        """Column-parallel gate_up channels: multiples of 32 (4 tiles of 8 up + 8 gate rows: one 2-D workgroup)."""
        return shard_range(I, self.world, self.rank, 32)

    def vocab_range(self, V: int) -> Tuple[int, int]:
        return shard_range(V, self.world, self.rank, 16)

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return t
        return self.comm.all_reduce(t)

    def _all_gather_columns(self, full: torch.Tensor, n: int, align: int, key: str) -> torch.Tensor:
        """full [T, n]: every rank has written its own column range (shard_range(n, world, rank, align)); on return
        every rank holds all columns.  Moves 1/world of what an all-reduce of the zero-padded tensor would."""
        T = full.shape[0]
        ranges = [shard_range(n, self.world, r, align) for r in range(self.world)]
        wmax = max(hi - lo for lo, hi in ranges)
        lo, hi = ranges[self.rank]
        send = self._buf(key + ".send", (T, wmax), full.dtype, full.device)
        recv = self._buf(key + ".recv", (self.world, T, wmax), full.dtype, full.device)
        send[:, :hi - lo].copy_(full[:, lo:hi])
        self.comm.all_gather(recv.view(self.world * T, wmax), send)
        if all(h - l == wmax for l, h in ranges):      # equal shards: one strided copy
            full.view(T, self.world, wmax).copy_(recv.permute(1, 0, 2))
        else:
            for r, (l, h) in enumerate(ranges):
                full[:, l:h].copy_(recv[r, :, :h - l])
        return full

    def all_gather_channels(self, act: torch.Tensor, I: int) -> torch.Tensor:
        """Column-parallel gate_up: act [T, I] with this rank's channel_range written -> complete on every rank."""
        if self.world == 1:
            return act
        return self._all_gather_columns(act, I, 32, "channels")

    def all_gather_vocab(self, local: torch.Tensor, out: torch.Tensor, V: int) -> torch.Tensor:
        """local [T, V_r] (this rank's vocab range) -> out [T, V] (logits_processor.py:104-107)."""
        if self.world == 1:
            out.copy_(local)
            return out
        lo, hi = self.vocab_range(V)
        out[:, lo:hi].copy_(local)
        return self._all_gather_columns(out, V, 16, "vocab")

    def broadcast_object(self, obj, src: int = 0):
        """Control plane (spec_decode_worker.py:524-538: broadcast_tensor_dict of the per-step control scalars)."""
        if self.world == 1:
            return obj
        return self.comm.broadcast_object(obj, src)


def init_from_env(device: str) -> TensorParallel:
    """torchrun environment -> process group (RCCL).  MASTER_ADDR must be 127.0.0.1 on the single-node boxes."""
    import os
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=torch.device(device))
    return TensorParallel(rank, world, None)


def measure_collective_us(tp: "TensorParallel", T: int, H: int, I: int, device, iters: int = 20) -> dict:
    """Time the three per-layer collectives of the sharded verify pass on THIS job's communicator, as they run in the
    cycle (same shapes, same stream, back to back inside a captured graph where the backend allows it): the fp32
    all-reduce of a [T, H] row-parallel partial (o_proj, down_proj) and the all-gather of the column-parallel gate_up
    channels.  MAX over ranks, so that every rank plans from the same numbers.  Returns microseconds per collective."""
    dev = torch.device(device)
    part = torch.zeros(T, H, dtype=torch.float32, device=dev)
    act = torch.zeros(T, I, dtype=torch.float16, device=dev)

    def body():
        tp.all_reduce(part)
        tp.all_gather_channels(act, I)
        tp.all_reduce(part)

    def timed(fn, n):
        if dev.type != "cuda":
            import time
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            return (time.perf_counter() - t0) / n * 1e6
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b) * 1e3 / n

    body()
    mode, fn = "eager", body
    on_gpu = dev.type == "cuda" and not tp.backend.endswith("gloo") and tp.backend != "threads"
    if on_gpu:
        g = None
        try:   # the cycle replays its collectives from a hipGraph: measure them the same way
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                body()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
        except Exception:
            torch.cuda.synchronize(dev)
            g = None
        # The mode decides how many collectives a rank issues from here on: it must be the SAME on every rank, or the
        # ranks deadlock on mismatched collectives.  A capture records, it does not execute -- so nothing has been
        # issued yet; agree first (MIN over the ranks of "my capture worked"), replay only if everybody can.
        if agree_all(tp, g is not None, dev):
            g.replay()
            mode, fn = "graph", g.replay
    per3 = timed(fn, iters)
    ar = timed((lambda: tp.all_reduce(part)), iters)     # on every rank, in every mode: one sequence of collectives
    t = torch.tensor([per3, ar], dtype=torch.float64, device=dev if on_gpu else "cpu")
    if tp.backend != "threads" and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=tp.group)
    per3 = float(t[0])
    out = {"three_collectives_per_layer_us": round(per3, 2), "per_collective_us": round(per3 / 3.0, 2), "mode": mode,
           "backend": tp.backend, "all_reduce_f32_shape": [T, H], "all_gather_f16_shape": [T, I // tp.world]}
    out["all_reduce_alone_us"] = round(float(t[1]), 2)
    return out


def agree_all(tp, ok: bool, dev) -> bool:
    """True iff `ok` holds on EVERY rank of tp's group (one small MIN all-reduce; ThreadComm / single rank: local)."""
    if tp.backend == "threads" or not dist.is_initialized() or tp.world <= 1:
        return bool(ok)
    on_gpu = torch.device(dev).type == "cuda" and not tp.backend.endswith("gloo")
    f = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if on_gpu else "cpu")
    dist.all_reduce(f, op=dist.ReduceOp.MIN, group=tp.group)
    return bool(int(f.item()))


def measure_vocab_gather_us(tp: "TensorParallel", B: int, V: int, device, iters: int = 20) -> float:
    """Time the draft pass's logits exchange -- all_gather_vocab of a [B, V] fp16 matrix, copies included -- on THIS job's
    communicator (captured where the backend allows, all ranks agreeing on the mode), MAX over ranks, microseconds."""
    dev = torch.device(device)
    lo, hi = tp.vocab_range(V)
    local = torch.zeros(B, hi - lo, dtype=torch.float16, device=dev)
    out = torch.zeros(B, V, dtype=torch.float16, device=dev)

    def body():
        tp.all_gather_vocab(local, out, V)
    body()
    on_gpu = dev.type == "cuda" and not tp.backend.endswith("gloo") and tp.backend != "threads"
    fn = body
    if on_gpu:
        g = None
        try:
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                body()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
        except Exception:
            torch.cuda.synchronize(dev)
            g = None
        if agree_all(tp, g is not None, dev):
            g.replay()
            fn = g.replay
    import time
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    us = (time.perf_counter() - t0) / iters * 1e6
    t = torch.tensor([us], dtype=torch.float64, device=dev if on_gpu else "cpu")
    if tp.backend != "threads" and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=tp.group)
    return float(t[0])


def shard_draft_vocab_pays(lm_head_bytes: int, world: int, gather_us: float) -> bool:
    """Vocab-parallel lm_head on the draft pass saves (1 - 1/world) of the lm_head stream per forward and costs the logits
    all-gather (its copies and launch boundaries inside the measurement)."""
    return lm_head_bytes * (1.0 - 1.0 / world) / LM_HEAD_BYTES_PER_US > gather_us


def shard_layers_pays_measured(layer_weight_bytes: int, world: int, three_collectives_us: float) -> bool:
    """The plan from MEASURED collective cost: sharding saves (1 - 1/world) of the layer's weight stream and costs the
    three collectives (plus their launch boundaries, already inside the measurement)."""
    saved = layer_weight_bytes * (1.0 - 1.0 / world) / STREAM_BYTES_PER_US
    return saved > three_collectives_us


def attach_tp(m, rank: int, world: int, tokens: Optional[int] = None, group=None, draft_tokens: Optional[int] = None):
    """Attach the tensor-parallel context to a model every rank holds in full.  The plan (shard the decoder layers of the
    verify pass or keep them replicated) comes from the collectives' cost MEASURED on this job's communicator when
    `tokens` (the verify pass's T) is given; QSPEC_TP_LAYERS=0/1 forces either plan; without a measurement the
    planning constants decide."""
    import os
    cfg = m.config
    layer_bytes = sum(lin.weight.numel() for lin in m.layers[0].linears())
    m.tp = TensorParallel(rank, world, group, shard_layers=shard_layers_pays(layer_bytes, world))
    m.tp.plan_basis = "forced by QSPEC_TP_LAYERS" if os.environ.get("QSPEC_TP_LAYERS") is not None else \
        "planning constants (parallel.COLLECTIVE_US): no measurement requested"
    m.tp.collective_us = None
    if world > 1 and tokens is not None:
        m.tp.collective_us = measure_collective_us(m.tp, tokens, cfg.hidden_size, cfg.intermediate_size, m.device)
        if os.environ.get("QSPEC_TP_LAYERS") is None:
            m.tp.shard_layers = shard_layers_pays_measured(layer_bytes, world,
                                                           m.tp.collective_us["three_collectives_per_layer_us"])
            m.tp.plan_basis = "collective cost measured in this job (parallel.measure_collective_us)"
    # draft pass: vocab-parallel lm_head + logits all-gather, decided the same way (draft_tokens = the batch size)
    forced = os.environ.get("QSPEC_TP_DRAFT_VOCAB")
    m.tp.vocab_gather_us = None
    if forced is not None:
        m.tp.shard_draft_vocab = world > 1 and forced != "0"
    elif world > 1 and draft_tokens is not None:
        m.tp.vocab_gather_us = measure_vocab_gather_us(m.tp, draft_tokens, cfg.vocab_size, m.device)
        m.tp.shard_draft_vocab = shard_draft_vocab_pays(m.lm_head.numel() * 2, world, m.tp.vocab_gather_us)
    return m


def build_tp_model(cfg, device: str, world: int, rank: int, seed: int = 0, lm_head_std: float = 0.02,
                   tokens: Optional[int] = None, draft_tokens: Optional[int] = None):
    """Same synthetic weights on every rank (same seed), TP context attached (attach_tp)."""
    from .model import QuarotLlamaForCausalLM
    m = QuarotLlamaForCausalLM(cfg, device).init_synthetic(seed, lm_head_std)
    return attach_tp(m, rank, world, tokens, draft_tokens=draft_tokens)
