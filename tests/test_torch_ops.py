"""`torch.ops.qspec.*` (qspec_amd/torch_ops.py): the reference's operator names behind the PyTorch dispatcher."""
import numpy as np
import pytest
import torch


def test_ops_are_registered_and_have_no_cpu_kernel():
    import qspec_amd.torch_ops as t
    for name in t.OPS:
        assert hasattr(torch.ops.qspec, name), name
    with pytest.raises(NotImplementedError):       # the product path fails loudly without the HIP backend
        torch.ops.qspec.fast_hadamard_transform(torch.zeros(2, 4, dtype=torch.float16), 1.0)
    with pytest.raises(NotImplementedError):
        torch.ops.qspec.rowwise_scaled_linear_cutlass_s4s4_unified(
            torch.zeros(1, 64, dtype=torch.int8), torch.ones(1, dtype=torch.float16), torch.zeros(16, 64, dtype=torch.int8),
            torch.ones(16, dtype=torch.float16), None, torch.empty(1, 16, dtype=torch.float16))


@pytest.mark.gpu
def test_dispatcher_ops_equal_the_direct_entries(oracle):
    import qspec_amd.torch_ops  # noqa: F401
    from qspec_amd import ops
    dev = "cuda:0"
    rng = np.random.default_rng(3)
    M, N, K = 4, 128, 4096
    xq = torch.from_numpy(rng.integers(-128, 128, (M, K // 2)).astype(np.int8)).to(dev)
    xs = torch.from_numpy((rng.random(M) * 0.1 + 0.01).astype(np.float16)).to(dev)
    wq = torch.from_numpy(rng.integers(-128, 128, (N, K // 2)).astype(np.int8)).to(dev)
    ws = torch.from_numpy((rng.random(N) * 0.01 + 0.001).astype(np.float16)).to(dev)
    a = torch.empty(M, N, dtype=torch.float16, device=dev); b = torch.empty_like(a)
    r = torch.ops.qspec.rowwise_scaled_linear_cutlass_s4s4_unified(xq, xs, wq, ws, None, a)
    ops.rowwise_scaled_linear_cutlass_s4s4_unified(xq, xs, wq, ws, None, b)
    assert r.data_ptr() == a.data_ptr() and torch.equal(a.view(torch.int16), b.view(torch.int16))
    x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float16)).to(dev)
    torch.ops.qspec.w4a16_matmul(x, wq, a, ws.view(N, 1))
    ops.w4a16_linear(x, wq, ws, b)
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    q = torch.empty(M, K // 2, dtype=torch.int8, device=dev); s = torch.empty(M, dtype=torch.float16, device=dev)
    torch.ops.qspec.rms_norm_general_fuse_sum_i4(q, x, None, s, 1e-5, True)
    q0, s0, _ = oracle.ln_quant_i4(x.cpu().numpy(), 1e-5)
    assert np.array_equal(q.cpu().numpy(), q0) and np.array_equal(s.cpu().numpy().view(np.uint16), s0.view(np.uint16))
    y = torch.ops.qspec.fast_hadamard_transform(x.view(-1, 512), 0.0442)
    assert np.array_equal(y.cpu().numpy().view(np.uint16), oracle.fwht(x.cpu().numpy().reshape(-1, 512), 0.0442).view(np.uint16))
