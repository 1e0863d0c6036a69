import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qspec_amd import ops
dev = "cuda:0"
M, H, I = 4, 4096, 1792
hidden = torch.randn(M, H, device=dev).half(); delta = (torch.randn(M, H, device=dev) * 0.3).half(); hout = torch.empty_like(hidden)
w = torch.randint(-128, 127, (2 * I, H // 2), dtype=torch.int8, device=dev)
sc = torch.rand(2 * I, device=dev).half() * 0.01
act = torch.empty(M, I, dtype=torch.float16, device=dev)
print("launch", flush=True)
ops.ln_gate_up_silu_linear(hidden, delta, hout, 1e-5, w, sc, act)
torch.cuda.synchronize()
print("ok", act.float().abs().mean().item(), flush=True)
