"""Time the M-tiled W4A16 GEMM at prefill shapes (TFLOP/s); `--lib` adds dequant + library GEMM for comparison.
QSPEC_TILED_MT / QSPEC_TILED_S force the launch plan (sweeps run this script once per setting)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qspec_amd import ops

dev = "cuda:0"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

lib = "--lib" in sys.argv
Ms = [int(a) for a in sys.argv[1:] if a.isdigit()] or [64, 128, 192, 512, 2048]
tag = f"MT={os.environ.get('QSPEC_TILED_MT','-')} S={os.environ.get('QSPEC_TILED_S','-')}"
for M in Ms:
    for N, K in ((6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)):
        x = torch.randn(M, K, device=dev, dtype=torch.float16)
        wq = torch.randint(-128, 128, (N, K // 2), device=dev, dtype=torch.int8)
        ws = torch.full((N,), 0.001, device=dev, dtype=torch.float16)
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        us = t(lambda: ops.w4a16_linear(x, wq, ws, out))
        fl = 2.0 * M * N * K
        line = f"{tag} M={M:5d} N={N:6d} K={K:6d} tiled {us:8.1f} us {fl/us/1e6:7.1f} TF"
        if lib:
            wd = torch.empty(N, K, device=dev, dtype=torch.float16)
            def f():
                ops.dequant_w4(wq, ws, wd); torch.matmul(x, wd.t(), out=out)
            us2 = t(f)
            us3 = t(lambda: torch.matmul(x, wd.t(), out=out))
            line += f" | dequant+lib {us2:8.1f} us | lib only {us3:8.1f} us {fl/us3/1e6:7.1f} TF"
        print(line, flush=True)
if "--no-head" in sys.argv or os.environ.get("QSPEC_TILED_MT"): sys.exit(0)
# ---- lm_head (fp16 x fp16)
N, K = 128256, 4096
w = (torch.randn(N, K, device=dev) * 0.02).half()
for M in (4, 16, 32, 96, 192, 512):
    x = torch.randn(M, K, device=dev, dtype=torch.float16)
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    us = t(lambda: ops.linear_f16(x, w, out))
    print(f"lm_head M={M:4d}: {us:8.1f} us  {N*K*2/us/1e6:6.2f} TB/s weights  {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)
