"""Micro-benchmark of the weight-streaming GEMMs (achieved HBM GB/s per shape).  Dev tool."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qspec_amd import ops

def bench(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters): fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e3  # us

dev = "cuda:0"
shapes = [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)]
L = 8  # rotate over L weight copies so the stream does not sit in the 256 MiB infinity cache
for M in (4, 16, 32):
    for N, K in shapes:
        ws = [torch.randint(-128, 127, (N, K // 2), dtype=torch.int8, device=dev) for _ in range(L)]
        sc = torch.rand(N, device=dev).half() * 0.01
        xq = torch.randint(-128, 127, (M, K // 2), dtype=torch.int8, device=dev)
        xs = torch.rand(M, device=dev).half()
        x = torch.randn(M, K, device=dev).half()
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        i = [0]
        def f4():
            ops.rowwise_scaled_linear_cutlass_s4s4_unified(xq, xs, ws[i[0] % L], sc, None, out); i[0] += 1
        def f16():
            ops.w4a16_linear(x, ws[i[0] % L], sc, out); i[0] += 1
        t4, t16 = bench(f4), bench(f16)
        gb = N * K / 2 / 1e9
        print(f"M={M:3d} N={N:5d} K={K:5d}  w4a4 {t4:7.1f} us {gb/t4*1e6:7.0f} GB/s   w4a16 {t16:7.1f} us {gb/t16*1e6:7.0f} GB/s", flush=True)
        del ws
V, H = 128256, 4096
w = (torch.randn(V, H, device=dev) * 0.02).half()
for M in (4, 16):
    x = torch.randn(M, H, device=dev).half(); out = torch.empty(M, V, dtype=torch.float16, device=dev)
    t = bench(lambda: ops.linear_f16(x, w, out), iters=20)
    t2 = bench(lambda: torch.matmul(x, w.t()), iters=20)
    print(f"lm_head M={M}: ours {t:7.1f} us {V*H*2/1e9/t*1e6:7.0f} GB/s   torch.matmul {t2:7.1f} us {V*H*2/1e9/t2*1e6:7.0f} GB/s")
