"""Body of __graft_entry__.smoke(): imports the oracle, so it lives outside the product modules' import graph
(nothing in qspec_amd imports this file)."""
import numpy as np
import torch


def run():
    import oracle as O
    from qspec_amd import ops
    dev = "cuda:0"
    rng = np.random.default_rng(0)
    T, H, N = 4, 4096, 256
    x = rng.standard_normal((T, H)).astype(np.float16)
    w = O.pack_i4(rng.integers(-8, 8, (N, H)).astype(np.int8))
    ws = (rng.random(N) * 0.01 + 0.001).astype(np.float16)
    q0, s0, _ = O.ln_quant_i4(x, 1e-5)
    ref = O.gemm_w4a4(q0, s0, w, ws)
    xd = torch.from_numpy(x).to(dev)
    q = torch.empty(T, H // 2, dtype=torch.int8, device=dev)
    s = torch.empty(T, dtype=torch.float16, device=dev)
    ops.rms_norm_general_fuse_sum_i4(q, xd, None, s, 1e-5)
    out = torch.empty(T, N, dtype=torch.float16, device=dev)
    ops.rowwise_scaled_linear_cutlass_s4s4_unified(q, s, torch.from_numpy(w).to(dev), torch.from_numpy(ws).to(dev), None, out)
    torch.cuda.synchronize()
    assert np.array_equal(q.cpu().numpy(), q0), "LN+int4 quant mismatch"
    assert np.array_equal(out.cpu().numpy().view(np.uint16), ref.view(np.uint16)), "W4A4 GEMM mismatch"
    print("smoke ok: LN+int4 quant and W4A4 GEMM bit-exact vs oracle")
