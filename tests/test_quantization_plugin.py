"""The vLLM quantization-plugin mirror (qspec_amd/quantization.py): config surface and parameter registration on the
CPU; `apply` against the operator entries on the GPU."""
import numpy as np
import pytest
import torch


def test_config_surface_and_create_weights():
    from qspec_amd.quantization import QSpecConfig, QSpecLinearMethod
    cfg = QSpecConfig.from_config({"w_bits": 4, "a_bits": 4, "a_clip_ratio": 0.9})
    assert cfg.get_name() == "qspec" and cfg.get_supported_act_dtypes() == [torch.float16]
    assert isinstance(cfg.get_min_capability(), int) and cfg.get_config_filenames()
    assert cfg.clip_ratio == 0.9
    with pytest.raises(ValueError):
        QSpecConfig.from_config({"w_bits": 8})
    layer = torch.nn.Module()
    assert cfg.get_quant_method(layer, "model.embed_tokens") is None
    assert cfg.get_quant_method(layer, "lm_head") is None
    method = cfg.get_quant_method(layer, "model.layers.0.self_attn.qkv_proj")
    assert isinstance(method, QSpecLinearMethod)
    seen = {}
    method.create_weights(layer, 4096, [4096, 1024, 1024], 4096, 6144, torch.float16,
                          weight_loader=lambda p, w: seen.setdefault("called", True))
    assert layer.weight.shape == (6144, 2048) and layer.weight.dtype == torch.int8 and not layer.weight.requires_grad
    assert layer.weight_scales.shape == (6144, 1) and layer.weight_scales.dtype == torch.float16
    assert layer.weight.input_dim == 1 and layer.weight.output_dim == 0 and layer.weight.pack_factor == 2
    assert callable(layer.weight.weight_loader) and layer.logical_widths == [4096, 1024, 1024]
    assert set(dict(layer.named_parameters())) == {"weight", "weight_scales"}    # the reference checkpoint's names
    with pytest.raises(ValueError):
        method.create_weights(torch.nn.Module(), 100, [64], 100, 64, torch.float16)
    with pytest.raises(ValueError):
        method.create_weights(torch.nn.Module(), 128, [64], 128, 64, torch.bfloat16)


@pytest.mark.gpu
def test_apply_runs_both_views_of_one_buffer(oracle):
    from qspec_amd import ops
    from qspec_amd.quantization import QSpecConfig
    from qspec_amd.quarot_nn import PackedQuantizedTensor
    dev = "cuda:0"
    rng = np.random.default_rng(0)
    M, N, K = 4, 256, 4096
    layer = torch.nn.Module()
    method = QSpecConfig().get_quant_method(layer, "model.layers.0.mlp.down_proj")
    method.create_weights(layer, K, [N], K, N, torch.float16)
    layer.to(dev)
    w = rng.integers(-128, 128, (N, K // 2)).astype(np.int8)
    ws = (rng.random((N, 1)) * 0.01 + 0.001).astype(np.float16)
    layer.weight.data.copy_(torch.from_numpy(w))
    layer.weight_scales.data.copy_(torch.from_numpy(ws))
    ptr = layer.weight.data_ptr()
    # draft view
    xq = rng.integers(-128, 128, (M, K // 2)).astype(np.int8)
    xs = (rng.random(M) * 0.1 + 0.01).astype(np.float16)
    y4 = method.apply(layer, PackedQuantizedTensor(torch.from_numpy(xq).to(dev), torch.from_numpy(xs).to(dev)))
    torch.cuda.synchronize()
    assert np.array_equal(y4.cpu().numpy().view(np.uint16), oracle.gemm_w4a4(xq, xs, w, ws.reshape(-1)).view(np.uint16))
    # verify view, batched leading dims
    x = (rng.standard_normal((2, 3, K))).astype(np.float16)
    y16 = method.apply(layer, torch.from_numpy(x).to(dev))
    torch.cuda.synchronize()
    ref = oracle.gemm_w4a16(x.reshape(-1, K), w, ws.reshape(-1)).reshape(2, 3, N)
    assert y16.shape == (2, 3, N)
    assert np.abs(y16.cpu().numpy().astype(np.float64) - ref.astype(np.float64)).max() <= 1e-3 * max(1.0, np.abs(ref).max())
    assert layer.weight.data_ptr() == ptr    # neither view copies or rewrites the buffer
