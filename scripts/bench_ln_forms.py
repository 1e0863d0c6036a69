"""Dev tool: the draft layer's norm + GEMM pairs, fused (norm in the GEMM prologue; hand-off from M = 8) against a
separate norm launch + the (xq, xs) GEMM, on cold weights, replayed from a graph.  us per PAIR."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qspec_amd import ops
dev = "cuda:0"
H, I, NQ, NKV, D = 4096, 14336, 32, 8, 128
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
cs = torch.randn(8192, 128, device=dev).half()
kc = torch.zeros(256, 16, NKV, D, dtype=torch.float16, device=dev); vc = torch.zeros_like(kc)
for M in (4, 8, 16, 32):
    hid = torch.randn(M, H, device=dev).half()
    pos = torch.randint(0, 4096, (M,), device=dev); slots = torch.arange(M, device=dev, dtype=torch.int64)
    xq = torch.empty(M, H // 2, dtype=torch.int8, device=dev); xs = torch.empty(M, dtype=torch.float16, device=dev)
    for name, N in (("qkv", (NQ + 2 * NKV) * D), ("gate_up", 2 * I)):
        L = max(2, int(600e6 // (N * H // 2)))
        ws = [torch.randint(-128, 127, (N, H // 2), dtype=torch.int8, device=dev) for _ in range(L)]
        sc = torch.rand(N, device=dev).half() * 0.01
        qkv = torch.empty(M, N, dtype=torch.float16, device=dev); act = torch.empty(M, I, dtype=torch.float16, device=dev)
        if name == "qkv":
            fused = lambda w: (lambda: ops.ln_qkv_rope_linear(hid, None, None, 1e-5, w, sc, qkv, pos, cs, kc, vc, slots, NQ, NKV, D))
            def sep(w):
                def f():
                    ops.add_rms_norm_i4(xq, xs, None, hid, None, 1e-5)
                    ops.qkv_rope_linear(xq, xs, w, sc, qkv, pos, cs, kc, vc, slots, NQ, NKV, D)
                return f
        else:
            fused = lambda w: (lambda: ops.ln_gate_up_silu_linear(hid, None, None, 1e-5, w, sc, act))
            def sep(w):
                def f():
                    ops.add_rms_norm_i4(xq, xs, None, hid, None, 1e-5)
                    ops.gate_up_silu_linear(xq, xs, w, sc, act)
                return f
        t_sep = timeit([sep(w) for w in ws])
        t_fused = timeit([fused(w) for w in ws]) if ops.ln_linear_s4s4_supported(M, N, H) else float("nan")
        print(f"M={M:2d} {name:8s} separate norm + GEMM {t_sep:6.2f} us | fused {t_fused:6.2f} us", flush=True)
        del ws
