"""Dev tool: phase stamps of the LN-prologue W4A4 GEMM (needs the -DQS_STREAM_STAMPS build via QSPEC_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qspec_amd import ops
dev = "cuda:0"
M, H, I = int(os.environ.get("M", 4)), 4096, 14336
hidden = torch.randn(M, H, device=dev).half(); delta = (torch.randn(M, H, device=dev) * 0.3).half(); hout = torch.empty_like(hidden)
ws = [torch.randint(-128, 127, (2 * I, H // 2), dtype=torch.int8, device=dev) for _ in range(6)]
sc = torch.rand(2 * I, device=dev).half() * 0.01
act = torch.empty(M, I, dtype=torch.float16, device=dev)
for w in ws:
    ops.ln_gate_up_silu_linear(hidden, delta, hout, 1e-5, w, sc, act)
torch.cuda.synchronize()
st = hout.view(-1)[-32:].view(torch.int64).cpu().tolist()
print("cycles [load issue, LN compute, main loop, last compute, last finish]:", [st[i + 1] - st[i] for i in range(5)])
ln = hout.view(-1)[-64:-32].view(torch.int64).cpu().tolist()
print("  inside the norm, from the kernel's first stamp: [residual stream arrived, mean tree done, variance tree done, "
      "quantised, barrier passed]:", [ln[i] - st[0] for i in range(5)])
# the form the draft pass launches (no delta, no write-back): stamps through qspec_debug_stamps()
import ctypes
for w in ws:
    ops.ln_gate_up_silu_linear(hidden, None, None, 1e-5, w, sc, act)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 8)()
lib = ctypes.CDLL(os.environ["QSPEC_HIP_LIB"])
assert lib.qspec_debug_stamps(buf) == 0
st = list(buf)
print("LN1 form, cycles [requests issued, norm joined / norm done, main loop, last compute, last finish]:", [st[i + 1] - st[i] for i in range(5)])
span = (ctypes.c_longlong * 2048)()
assert lib.qspec_debug_wgspan(span) == 0
import numpy as np
sp = np.array(list(span), dtype=np.int64).reshape(1024, 2)[:256]
t0 = sp[:, 0].min()
st_, en_ = (sp[:, 0] - t0) * 0.01, (sp[:, 1] - t0) * 0.01      # us (100 MHz)
print("per workgroup (us from the first workgroup's start): starts min/median/max %.2f %.2f %.2f | ends min/10%%/median/90%%/max %.2f %.2f %.2f %.2f %.2f"
      % (st_.min(), np.median(st_), st_.max(), en_.min(), np.quantile(en_, 0.1), np.median(en_), np.quantile(en_, 0.9), en_.max()))
ln8 = (ctypes.c_longlong * 8)()
assert lib.qspec_debug_lnst(ln8) == 0
l8 = list(ln8)
print("norm wave 0 (LN1S), cycles from the kernel's first stamp: [norm code entered, row arrived, mean known, variance known, row quantised]:", [l8[i] - st[0] for i in range(5)])
