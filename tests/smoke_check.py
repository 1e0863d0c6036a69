"""Body of __graft_entry__.smoke(): one small invocation of the hot path on cuda:0 checked against the CPU oracle.
It imports the oracle, so it lives under tests/ (the package `qspec_amd` never touches `oracle/`)."""
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
    # the draft pass's fused launch: residual add + LN + int4 quant + gate_up GEMM + silu(gate)*up, vs the oracle's ops
    I = 128
    d = (rng.standard_normal((T, H)) * 0.3).astype(np.float16)
    wg = O.pack_i4(rng.integers(-8, 8, (2 * I, H)).astype(np.int8))
    wgs = (rng.random(2 * I) * 0.01 + 0.001).astype(np.float16)
    h0 = O.add_f16(x, d)
    q1, s1, _ = O.ln_quant_i4(h0, 1e-5)
    act0 = O.silu_mul(O.gemm_w4a4(q1, s1, wg, wgs), I)
    hout = torch.empty(T, H, dtype=torch.float16, device=dev)
    act = torch.empty(T, I, dtype=torch.float16, device=dev)
    ops.ln_gate_up_silu_linear(xd, torch.from_numpy(d).to(dev), hout, 1e-5, torch.from_numpy(wg).to(dev),
                               torch.from_numpy(wgs).to(dev), act)
    torch.cuda.synchronize()
    assert np.array_equal(hout.cpu().numpy().view(np.uint16), h0.view(np.uint16)), "residual add mismatch"
    assert np.array_equal(act.cpu().numpy().view(np.uint16), act0.view(np.uint16)), "fused LN + gate_up + silu mismatch"
    print("smoke ok: LN+int4 quant, W4A4 GEMM and the fused LN->gate_up->silu launch bit-exact vs oracle")
