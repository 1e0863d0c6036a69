"""Graph-replay micro-benchmark of the draft pass's W4A4 launches, cold weights (rotating over > 256 MiB of copies).
Dev tool.  QSPEC_OLD_GEMM=1 / QSPEC_STREAM_CAP=n select variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qspec_amd import ops
dev = "cuda:0"
ROT = float(os.environ.get("QSPEC_ROT_BYTES", "600e6"))   # bytes of weight copies rotated over (> 256 MiB = HBM-cold)
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
Ms = [int(m) for m in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["4"])]
H, I, nq, nkv, d = 4096, 14336, 32, 8, 128
for M in Ms:
    hidden = torch.randn(M, H, device=dev).half(); delta = (torch.randn(M, H, device=dev) * 0.3).half(); hout = torch.empty_like(hidden)
    cs = torch.randn(8192, 128, device=dev).half(); pos = torch.randint(0, 8192, (M,), device=dev)
    kc = torch.zeros(256, 16, nkv, d, device=dev, dtype=torch.float16); vc = torch.zeros_like(kc)
    slots = torch.arange(M, device=dev, dtype=torch.int64)
    for name, N, K in (("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)):
        L = max(2, int(ROT // (N * K // 2)))
        ws = [torch.randint(-128, 127, (N, K // 2), dtype=torch.int8, device=dev) for _ in range(L)]
        sc = torch.rand(N, device=dev).half() * 0.01
        xq = torch.randint(-128, 127, (M, K // 2), dtype=torch.int8, device=dev); xs = torch.rand(M, device=dev).half()
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        act = torch.empty(M, I, dtype=torch.float16, device=dev)
        gb = N * K / 2 / 1e9
        res = {}
        if name == "qkv":
            res["q"] = timeit([(lambda w=w: ops.qkv_rope_linear(xq, xs, w, sc, out, pos, cs, kc, vc, slots, nq, nkv, d)) for w in ws])
            res["ln"] = timeit([(lambda w=w: ops.ln_qkv_rope_linear(hidden, delta, hout, 1e-5, w, sc, out, pos, cs, kc, vc, slots, nq, nkv, d)) for w in ws])
        elif name == "gate_up":
            res["q"] = timeit([(lambda w=w: ops.gate_up_silu_linear(xq, xs, w, sc, act)) for w in ws])
            res["ln"] = timeit([(lambda w=w: ops.ln_gate_up_silu_linear(hidden, delta, hout, 1e-5, w, sc, act)) for w in ws])
        else:
            res["q"] = timeit([(lambda w=w: ops.rowwise_scaled_linear_cutlass_s4s4_unified(xq, xs, w, sc, None, out)) for w in ws])
        print(f"M={M:2d} {name:8s} " + "  ".join(f"{k}: {v:6.2f} us ({gb/v*1e6:5.0f} GB/s)" for k, v in res.items()), flush=True)
        del ws
# ---- W4A16 (verify pass) at M = 16
M = 16
x = torch.randn(M, 14336, device=dev).half()
cs = torch.randn(8192, 128, device=dev).half(); pos = torch.randint(0, 8192, (M,), device=dev)
kc = torch.zeros(256, 16, nkv, d, device=dev, dtype=torch.float16); vc = torch.zeros_like(kc)
slots = torch.arange(M, device=dev, dtype=torch.int64)
for name, N, K in (("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)):
    L = max(2, int(ROT // (N * K // 2)))
    ws = [torch.randint(-128, 127, (N, K // 2), dtype=torch.int8, device=dev) for _ in range(L)]
    sc = torch.rand(N, device=dev).half() * 0.01
    xx = x[:, :K].contiguous()
    out = torch.empty(M, N, dtype=torch.float16, device=dev); act = torch.empty(M, I, dtype=torch.float16, device=dev)
    if name == "qkv":
        t = timeit([(lambda w=w: ops.qkv_rope_linear(xx, None, w, sc, out, pos, cs, kc, vc, slots, nq, nkv, d)) for w in ws])
    elif name == "gate_up":
        t = timeit([(lambda w=w: ops.gate_up_silu_linear(xx, None, w, sc, act)) for w in ws])
    else:
        t = timeit([(lambda w=w: ops.w4a16_linear(xx, w, sc, out)) for w in ws])
    print(f"W4A16 M=16 {name:8s} {t:6.2f} us ({N * K / 2 / 1e9 / t * 1e6:5.0f} GB/s)", flush=True)
    del ws
