"""Run under torchrun (2 ranks, gloo for the handle exchange, BOTH on cuda:0): the one-shot push all-reduce over
IPC-mapped peer buffers (qspec_amd/csrc/comm.hip) against the plain sum.  Prints ONESHOT_OK on rank 0.

Two processes on one GPU stand in for two GPUs: the peer buffer is reached through hipIpcOpenMemHandle exactly as it
would be over xGMI, and the two ranks' kernels wait for each other's flags while both are resident.  What this cannot
show is xGMI timing: the path stays opt-in (QSPEC_ONESHOT_AR=1) until it has been measured on a multi-GPU box."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from qspec_amd.parallel import OneShotComm
    comm = OneShotComm(rank, world, None, max_bytes=1 << 20)
    assert comm.backend == "oneshot+gloo"
    dev = "cuda:0"
    ok = True
    for it, (T, H) in enumerate([(16, 4096), (4, 4096), (32, 8192), (16, 4096), (3, 1024), (16, 4096)]):
        g = torch.Generator().manual_seed(100 * it)            # every rank can rebuild every rank's input
        parts = [(torch.randn(T, H, generator=g) * (r + 1)).float() for r in range(world)]
        x = parts[rank].to(dev)
        comm.all_reduce(x)
        torch.cuda.synchronize()
        ref = parts[0].clone()
        for r in range(1, world):
            ref = ref + parts[r]                               # rank order, fp32: the kernel's own order
        ok = ok and torch.equal(x.cpu(), ref)
    # back-to-back calls inside ONE captured graph (device-side generation tags, no host counter)
    T, H = 16, 4096
    bufs = [torch.full((T, H), float(rank + 1 + i), dtype=torch.float32, device=dev) for i in range(6)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        comm.all_reduce(bufs[0].clone())
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    dist.barrier()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for b in bufs:
            comm.all_reduce(b)
    torch.cuda.synchronize()
    dist.barrier()
    for rep in range(3):
        for i, b in enumerate(bufs):
            b.fill_(float(rank + 1 + i + rep))
        graph.replay()
        torch.cuda.synchronize()
        for i, b in enumerate(bufs):
            want = float(sum(r + 1 + i + rep for r in range(world)))
            ok = ok and bool((b == want).all())
    # a message the one-shot path does not take (fp16) falls back to the library collective
    h = torch.full((8, 64), float(rank + 1), dtype=torch.float16, device=dev)
    comm.all_reduce(h)
    ok = ok and bool((h == float(sum(range(1, world + 1)))).all())
    err = comm.error()
    flags = [None] * world
    dist.all_gather_object(flags, (ok, err))
    if rank == 0:
        assert all(f == (True, 0) for f in flags), flags
        print("ONESHOT_OK world=%d" % world, flush=True)
    dist.barrier()
    # A LOST peer: rank 1 stops calling; rank 0's sticky error word is raised (as its first timed-out wait
    # would) and its next all-reduces must return at once instead of spinning ~20 s each.  Rank 1 keeps its buffer mapped.
    if rank == 0:
        import ctypes
        import time
        from qspec_amd import ops
        hip = ctypes.CDLL("libamdhip64.so")
        one = ctypes.c_uint32(1)
        assert hip.hipMemcpy(ctypes.c_void_p(comm.error_word_address()), ctypes.byref(one), 4, 1) == 0
        x = torch.ones(16, 4096, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            comm.all_reduce(x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert comm.error() == 1 and dt < 2.0, f"six all-reduces behind a raised error word took {dt:.1f} s"
        ops.collect_error_words([comm.error_word_address()], clear=True)
        torch.cuda.synchronize()
        assert comm.error() == 0
        print("ONESHOT_DEAD_PEER_OK %.3f s" % dt, flush=True)
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
