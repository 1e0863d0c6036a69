"""Dev tool: time the engine-form gate_up launch in a graph (cold weights) and, on a -DQS_ENG_STAMPS build
(QSPEC_HIP_LIB), print the phase stamps of workgroup 100's consumer wave 0 and loader 0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qspec_amd import ops
dev = "cuda:0"
M, H, I = int(os.environ.get("M", 4)), 4096, 14336
hidden = torch.randn(M, H, device=dev).half()
L = 10
ws = [torch.randint(-128, 127, (2 * I, H // 2), dtype=torch.int8, device=dev) for _ in range(L)]
sc = torch.rand(2 * I, device=dev).half() * 0.01
act = torch.empty(M, I, dtype=torch.float16, device=dev)
fs = [(lambda w: (lambda: ops.ln_gate_up_silu_linear(hidden, None, None, 1e-5, w, sc, act)))(w) for w in ws]
for f in fs: f()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(3):
        for f in fs: f()
g.replay(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): g.replay()
b.record(); torch.cuda.synchronize()
print(f"gate_up engine={os.environ.get('QSPEC_ENGINE','1')} lib={os.path.basename(os.environ.get('QSPEC_HIP_LIB','default'))}: {a.elapsed_time(b) / (15 * L) * 1e3:.2f} us per launch")
lib_path = os.environ.get("QSPEC_HIP_LIB")
if lib_path:
    import ctypes
    lib = ctypes.CDLL(lib_path)
    if hasattr(lib, "qspec_debug_engst"):
        buf = (ctypes.c_longlong * 384)()
        assert lib.qspec_debug_engst(buf) == 0
        st = list(buf)
        c, e = st[:64], st[320:]
        ld = [st[64 * (1 + i):64 * (2 + i)] for i in range(4)]
        t0 = min([c[0]] + [l[0] for l in ld])
        us = lambda v: (v - t0) / 100.0   # s_memrealtime: 100 MHz
        print("consumer wave 0 (us from start): start %.2f | rows requested %.2f | barrier passed %.2f | own row quantised %.2f | all rows %.2f | fragments %.2f"
              % tuple(us(c[i]) for i in range(6)))
        print("   fill in registers:", " ".join("%.2f" % us(c[6 + n]) for n in range(14)))
        if c[20] > t0:
            for n in range(10, 14):
                b = 20 + 5 * (n - 10)
                print("   fill %d: before confirm %.2f | in registers %.2f | next spec issued+waited %.2f | MFMAs issued %.2f | tile posted %s"
                      % (n, us(c[b]), us(c[6 + n]), us(c[b + 1]), us(c[b + 2]), ("%.2f" % us(c[b + 3])) if c[b + 3] > t0 else "-"))
        print("epilogue wave, tile done:", " ".join("%.2f" % us(e[n]) for n in range(7)))
        for i, l in enumerate(ld):
            k = [v for v in l[3:8] if v > t0]
            print("loader %d: start %.2f | own fills issued / drained:" % (i, us(l[1])), " ".join("%.2f" % us(v) for v in k))
