"""Graph-replay micro-benchmark of the paged attention kernel at the bench shapes (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qspec_amd import ops
dev = "cuda:0"
B, nq, nkv, d, bs = 4, 32, 8, 128, 16
ctx0, max_len = int(os.environ.get('CTX', 560)), int(os.environ.get('MAXLEN', 640))
for q_len in (1, 4):
    n_splits = (max_len + 127) // 128
    nb = B * (max_len // bs)
    kc = torch.randn(nb, bs, nkv, d, device=dev).half(); vc = torch.randn_like(kc)
    bt = torch.arange(nb, dtype=torch.int32, device=dev).view(B, -1).contiguous()
    T = B * q_len
    row = (nq + 2 * nkv) * d
    qkv = torch.randn(T, row, device=dev).half()
    ctx = torch.full((B,), ctx0, dtype=torch.int32, device=dev)
    qs = (torch.arange(B + 1, dtype=torch.int32, device=dev) * q_len).contiguous()
    ws = torch.zeros(ops.paged_attention_workspace_bytes(T, nq, d, n_splits), dtype=torch.uint8, device=dev)
    out = torch.empty(T, nq * d, dtype=torch.float16, device=dev)
    PART = os.environ.get('PARTIAL', '0') == '1'
    f = lambda: ops.paged_attention(qkv, row, kc, vc, bt, ctx, qs, T, q_len, nq, d ** -0.5, n_splits, ws, None if PART else out)
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    stx = ws[2048*4+9*8:2048*4+14*8].view(torch.int64).cpu().tolist()
    if any(stx): print('  [meta landed, bt issued, bt landed, data issued, data landed]:', stx)
    st = ws[2048*4:2048*4+9*8].view(torch.int64).cpu().tolist()
    if any(st): print('last-arriver WG cycle deltas [loads, QK, softmax, PV, store, handoff, acquire, merge]:', [st[i+1]-st[i] for i in range(8)])
    print(f"stage={os.environ.get('QS_ATT_STAGE','0')} q_len={q_len} n_splits={n_splits}: {a.elapsed_time(b)/250*1e3:.2f} us/launch", flush=True)
