"""Prompt pass (W4A16 prefill over the packed int4 weights) timing: one prompt of L tokens per sequence."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qspec_amd.model import CONFIGS, QuarotLlamaForCausalLM
from qspec_amd.spec_decode import QSpecEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cfg = CONFIGS["llama-3-8b"]
model = QuarotLlamaForCausalLM(cfg, "cuda:0").init_synthetic(0, 0.02)
g = torch.Generator().manual_seed(1)
prompts = [torch.randint(0, cfg.vocab_size, (L,), generator=g).tolist() for _ in range(B)]
for it in range(3):
    eng = QSpecEngine(model, 3, B, max_model_len=L + 64, block_size=16, max_new_tokens=16, use_graph=False, seed=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.add_sequences(prompts)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"prefill B={B} L={L}: {dt*1e3:.2f} ms, {B*L/dt:.0f} tok/s", flush=True)
    del eng
