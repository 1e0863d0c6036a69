"""Graph-replay micro-benchmark of the decode GEMMs: cold (rotating over copies > infinity cache) vs warm (same
weights every launch, resident in the 256 MiB infinity cache).  Dev tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qspec_amd import ops
dev = "cuda:0"
def timeit(fs, reps=3):
    for f in fs: f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            for f in fs: f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * reps * len(fs)) * 1e3
for M in (4, 16):
    for name, N, K in (("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)):
        L = max(2, int(600e6 // (N * K // 2)))
        ws = [torch.randint(-128, 127, (N, K // 2), dtype=torch.int8, device=dev) for _ in range(L)]
        sc = torch.rand(N, device=dev).half() * 0.01
        xq = torch.randint(-128, 127, (M, K // 2), dtype=torch.int8, device=dev); xs = torch.rand(M, device=dev).half()
        x = torch.randn(M, K, device=dev).half()
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        mk4 = lambda w: (lambda: ops.rowwise_scaled_linear_cutlass_s4s4_unified(xq, xs, w, sc, None, out))
        mk16 = lambda w: (lambda: ops.w4a16_linear(x, w, sc, out))
        gb = N * K / 2 / 1e9
        c4, w4 = timeit([mk4(w) for w in ws]), timeit([mk4(ws[0])] * 8)
        c16, w16 = timeit([mk16(w) for w in ws]), timeit([mk16(ws[0])] * 8)
        print(f"M={M:2d} {name:8s} w4a4 cold {c4:6.2f} us ({gb/c4*1e6:5.0f} GB/s) warm {w4:6.2f} us ({gb/w4*1e6:5.0f} GB/s) | "
              f"w4a16 cold {c16:6.2f} us ({gb/c16*1e6:5.0f} GB/s) warm {w16:6.2f} us ({gb/w16*1e6:5.0f} GB/s)", flush=True)
        del ws
