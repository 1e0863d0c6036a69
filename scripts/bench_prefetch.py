"""Dev tool: does a tile-aligned prefetch launch in front of a streaming GEMM leave the weights in the consumer's L2?
pair = prefetch_tiles(W) ; gemm(W) on rotating (cold) W; reported: gemm alone, prefetch alone, pair, pair - prefetch."""
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
M = 4
for name, N, K in (("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)):
    L = max(3, int(700e6 // (N * K // 2)))
    ws = [torch.randint(-128, 127, (N, K // 2), dtype=torch.int8, device=dev) for _ in range(L)]
    sc = torch.rand(N, device=dev).half() * 0.01
    xq = torch.randint(-128, 127, (M, K // 2), dtype=torch.int8, device=dev); xs = torch.rand(M, device=dev).half()
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    nt = N // 16
    for frac in (1.0, 0.5):
        npre = int(nt * frac)
        gemm = lambda w: (lambda: ops.rowwise_scaled_linear_cutlass_s4s4_unified(xq, xs, w, sc, None, out))
        pre = lambda w: (lambda: ops.prefetch_tiles(w, 0, npre, 256))
        def pair(w):
            def f():
                ops.prefetch_tiles(w, 0, npre, 256)
                ops.rowwise_scaled_linear_cutlass_s4s4_unified(xq, xs, w, sc, None, out)
            return f
        tg = timeit([gemm(w) for w in ws]); tp = timeit([pre(w) for w in ws]); tpair = timeit([pair(w) for w in ws])
        print(f"{name:8s} prefetch {frac:.1f}: gemm alone {tg:6.2f} us | prefetch alone {tp:6.2f} | pair {tpair:6.2f} | gemm behind prefetch {tpair - tp:6.2f}", flush=True)
    del ws
