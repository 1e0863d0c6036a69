"""Dev tool: phase stamps of the decode attention kernel (needs the -DQS_ATT_STAMPS build via QSPEC_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qspec_amd import ops
from qspec_amd.model import CONFIGS, QuarotLlamaForCausalLM
from qspec_amd.spec_decode import QSpecEngine
dev = "cuda:0"
cfg = CONFIGS["llama-3-8b"]
model = QuarotLlamaForCausalLM(cfg, dev).init_synthetic(0, 0.02)
B, ctx = 4, 512
eng = QSpecEngine(model, 3, B, max_model_len=ctx + 256, block_size=16, max_new_tokens=128, use_graph=False, seed=0)
g = torch.Generator(device=dev).manual_seed(1)
for kc, vc in eng.kv_caches:
    kc.copy_((torch.randn(kc.shape, generator=g, device=dev) * 0.5).half()); vc.copy_((torch.randn(vc.shape, generator=g, device=dev) * 0.5).half())
eng.seq_lens.fill_(ctx + 1); eng.gen_lens.fill_(1)
eng.last_token.copy_(torch.randint(0, cfg.vocab_size, (B,), generator=g, device=dev))
eng._len_ub = [ctx + 1] * B; eng._gen_ub = [1] * B; eng.n_active = B
eng.step(); torch.cuda.synchronize()
s, md = eng.scratch_draft, eng.md_draft
row = cfg.q_size + 2 * cfg.kv_size
for kc, vc in eng.kv_caches:      # 32 launches on 32 different caches: the last one's stamps are read
    ops.paged_attention(s.act_buffer_qkv[:B], row, kc, vc, md.block_tables, md.ctx_lens, md.q_start, B, md.max_q_len,
                        cfg.num_attention_heads, model.sm_scale, md.n_splits, s.attn_ws, None)
torch.cuda.synchronize()
st = s.attn_ws.view(torch.int32)[2048:2048 + 20].view(torch.int64).cpu().tolist()
print("attention (workgroup 0), ticks: entry -> metadata + table row here %d | -> K / V requested %d | -> K / V here %d | -> QK^T, scores + V in LDS %d | -> "
      "barrier + softmax + P in LDS %d | -> barrier %d | -> P.V done %d | -> partials stored %d | -> stores drained %d"
      % (st[0] - st[9], st[1] - st[0], st[2] - st[1], st[3] - st[2], st[7] - st[3], st[8] - st[7], st[4] - st[8], st[5] - st[4], st[6] - st[5]))
