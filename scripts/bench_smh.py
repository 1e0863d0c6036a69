"""Graph-replay micro-benchmark of mlp_hadamard / heads_hadamard / ln kernels (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qspec_amd import ops, hadamard_tables
dev = "cuda:0"
def timeit(f, n=50):
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * n) * 1e3
for T in (4, 16):
    I = 14336
    act = torch.randn(T, I, device=dev).half()
    had = hadamard_tables.get_hadK(I)[0].half().to(dev)
    q = torch.empty(T, I // 2, dtype=torch.int8, device=dev); sc = torch.zeros(1024, dtype=torch.float16, device=dev)
    o16 = torch.empty(T, I, dtype=torch.float16, device=dev)
    t0 = timeit(lambda: ops.mlp_hadamard(act, had, 28, 0.00835, q=q, scale=sc, workspace=None))
    print(f"T={T}: mlp_hadamard quant, one workgroup per token {t0:.2f} us", flush=True)
    t1 = timeit(lambda: ops.mlp_hadamard(act, had, 28, 0.00835, q=q, scale=sc))
    st = sc[64:64 + 20].view(torch.int64).cpu().tolist()
    t2 = timeit(lambda: ops.mlp_hadamard(act, had, 28, 0.00835, out_f16=o16))
    attn = torch.randn(T, 4096, device=dev).half(); q2 = torch.empty(T, 2048, dtype=torch.int8, device=dev)
    t3 = timeit(lambda: ops.heads_hadamard(attn, 0.1767, q=q2, scale=sc, heads=32))
    x = torch.randn(T, 4096, device=dev).half(); d = torch.randn(T, 4096, device=dev).half(); h = torch.empty_like(x)
    t4 = timeit(lambda: ops.add_rms_norm_i4(q2, sc, h, x, d, 1e-5))
    print(f"T={T}: mlp_hadamard quant {t1:.2f} us, fp16 {t2:.2f} us | heads_hadamard quant {t3:.2f} us | add_ln_quant {t4:.2f} us | stamps(barrier wait, mix, quant, ...) {st[:3]}", flush=True)
