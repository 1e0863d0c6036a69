"""The C-ABI library loads and exports exactly what include/qspec_hip.h declares (no GPU needed)."""
import ctypes
import os

import pytest

from qspec_amd import _lib


def test_library_is_built():
    assert os.path.exists(_lib.LIB_PATH), "run `python -c 'import __graft_entry__ as g; g.build()'`"


def test_header_and_binding_agree():
    assert _lib.header_symbols() == sorted(_lib.SIGNATURES)


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _lib.header_symbols():
        assert hasattr(lib, name), name


def test_load_binds_prototypes_and_reports_errors():
    lib = _lib.load()
    assert lib.qspec_abi_version() == 6
    # argument validation happens before any HIP call, so it is testable without a GPU
    rc = lib.qspec_rms_norm_general_fuse_sum_fp16(None, None, 1e-5, 4, 4096, None)
    assert rc != 0 and b"NULL" in lib.qspec_last_error()
    rc = lib.qspec_rowwise_scaled_linear_s4s4(1, 1, 1, 1, None, 1, 4, 100, 4096, None)
    assert rc != 0 and b"N % 16" in lib.qspec_last_error()
    with pytest.raises(_lib.QSpecLibraryError):
        _lib.check(rc, lib)


def test_ops_refuse_cpu_tensors():
    import torch
    from qspec_amd import ops
    x = torch.zeros(2, 4096, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.rms_norm_general_fuse_sum_fp16(torch.empty_like(x), x, 1e-5)
