"""Attention (partial mode) + merging head Hadamard over 32 different KV caches per graph (cold KV, as in the cycle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qspec_amd import ops
dev = "cuda:0"
B, nq, nkv, d, bs, L = int(os.environ.get("B", 4)), int(os.environ.get("NQ", 32)), int(os.environ.get("NKV", 8)), 128, 16, int(os.environ.get("L", 32))
ctx0 = int(os.environ.get("CTX", 512)); max_len = ctx0 + 128
for q_len in (1, 4):
    n_splits = int(os.environ.get("S", 8))
    nb = B * (max_len // bs)
    kcs = [torch.randn(nb, bs, nkv, d, device=dev).half() for _ in range(L)]
    vcs = [torch.randn(nb, bs, nkv, d, device=dev).half() for _ in range(L)]
    bt = torch.arange(nb, dtype=torch.int32, device=dev).view(B, -1).contiguous()
    T = B * q_len
    row = (nq + 2 * nkv) * d
    qkv = torch.randn(T, row, device=dev).half()
    ctx = torch.full((B,), ctx0, dtype=torch.int32, device=dev)
    qs = (torch.arange(B + 1, dtype=torch.int32, device=dev) * q_len).contiguous()
    ws = torch.zeros(ops.paged_attention_workspace_bytes(T, nq, d, n_splits), dtype=torch.uint8, device=dev)
    q = torch.empty(T, nq * d // 2, dtype=torch.int8, device=dev); sc = torch.empty(T, dtype=torch.float16, device=dev)
    def run(which):
        def body():
            for kc, vc in zip(kcs, vcs):
                if which != "had":
                    ops.paged_attention(qkv, row, kc, vc, bt, ctx, qs, T, q_len, nq, d ** -0.5, n_splits, ws, None)
                if which != "att":
                    ops.heads_hadamard_merged(ws, T, n_splits, T, nq, d, 0.1767, q=q, scale=sc)
        body(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): g.replay()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / (5 * L) * 1e3
    st = None
    t_att = run("att")
    stx = ws[2048 * 4:2048 * 4 + 9 * 8].view(torch.int64).cpu().tolist()
    stw = ws[8192:8192 + 25 * 8].view(torch.int64).cpu().tolist()
    if 0 < stw[0] <= 24:   # -DQS_ATT_STAMPS build of the waves kernel: wave 0 of workgroup (1, 1, 0)
        t = stw[1:1 + stw[0]]
        print("  waves-kernel stamps, ticks since the metadata arrived (per slice: arrived / QK + refills issued + V in LDS / slice done): "
              + " ".join(str(x - t[0]) for x in t[1:]), flush=True)
    if nq not in (32, 64):
        kvb = B * ctx0 * nkv * d * 2 * 2
        print(f"q_len={q_len}: attention(partials, cold KV) {t_att:.2f} us  ({kvb / t_att / 1e6:.2f} TB/s of KV)", flush=True)
        continue
    print(f"q_len={q_len}: attention(partials, cold KV) {t_att:.2f} us | merge+hadamard {run('had'):.2f} us | both {run('both'):.2f} us"
          + (f" | stamps {[stx[i+1]-stx[i] for i in range(5)]}" if any(stx) else ""), flush=True)
