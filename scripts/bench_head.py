"""lm_head micro-benchmark (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qspec_amd import ops
dev = "cuda:0"
V, H = 128256, 4096
w = (torch.randn(V, H, device=dev) * 0.02).half()
for M in (4, 16, 32):
    x = torch.randn(M, H, device=dev).half(); out = torch.empty(M, V, dtype=torch.float16, device=dev)
    ops.linear_f16(x, w, out); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4): ops.linear_f16(x, w, out)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 20 * 1e3
    ref = (x.float() @ w.float().t())
    err = (out.float() - ref).abs().max().item()
    print(f"M={M} lm_head {t:.1f} us ({V * H * 2 / t / 1e3:.0f} GB/s) max err {err:.3e}", flush=True)
