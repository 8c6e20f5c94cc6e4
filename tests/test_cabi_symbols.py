"""CPU: libinc_mi355x.so loads and exports exactly what include/inc_mi355x.h declares (no compute calls)."""

import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "inc_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(inc_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    from neural_compressor_amd import _lib

    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/inc_mi355x.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes bindings and header disagree"


def test_library_identity():
    from neural_compressor_amd import _lib

    assert _lib.lib.inc_abi_version() == _lib.ABI_VERSION
    assert _lib.lib.inc_target_arch() == b"gfx950"
    assert b"unsupported" in _lib.lib.inc_error_string(-2)


def test_bad_arguments_are_rejected_without_a_gpu():
    """Argument validation happens before any HIP call, so it is testable on a CPU-only box."""
    from neural_compressor_amd import _lib

    assert _lib.lib.inc_pack_rows(None, None, 0, 0, 4, 32, None) == -1
    assert _lib.lib.inc_woq_gemm(None, 2, None, None, None, None, None, None, 1, 1, 1, 1, 1, 4, None, 0, None) == -1
    assert _lib.lib.inc_woq_gemm_workspace_bytes(1, 4096, 4096) > 0
    assert _lib.lib.inc_woq_gemm_workspace_bytes(4096, 4096, 4096) == 0


def test_new_entry_points_validate_their_arguments():
    from neural_compressor_amd import _lib

    L = _lib.lib
    assert L.inc_w8a8_gemm(None, None, None, None, None, None, 2, 1, 1, 128, None, 0, None) == -1           # null pointers
    assert L.inc_w8a8_gemm_workspace_bytes(4096, 5120, 13824) > 0 and L.inc_w8a8_gemm_workspace_bytes(4096, 5120, 5120) == 0
    assert L.inc_w8a8_gemm_workspace_bytes(8192, 8192, 8192) == 0 and L.inc_w8a8_gemm_workspace_bytes(64, 64, 100) == 0
    assert L.inc_sq_quant_act(None, 2, 1, 16, 16, None, 1.0, 0.0, None, None) == -1
    assert L.inc_sq_channel_minmax(None, 2, 1, 1, 1, None, None, None) == -1
    assert L.inc_awq_repack(None, None, 8, 8, 1, 4, None, None, None) == -1
    assert L.inc_gptq_find_params_mse(None, 1, 1, 0, 1, 1, 4, 1, 100, 0.8, 2.4, None, None, 1, 0, None) == -1


def test_no_cpu_fallback():
    import pytest
    import torch

    from neural_compressor_amd import ops

    with pytest.raises(RuntimeError, match="HBM"):
        ops.pack_rows(torch.zeros(2, 8, dtype=torch.int32), 4, 32)
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear

    with pytest.raises(RuntimeError, match="HIP device"):
        MI355XWeightOnlyLinear(8, 8, device="cpu")
    from neural_compressor_amd.torch.algorithms.smooth_quant import W8A8Linear

    with pytest.raises(RuntimeError, match="HIP device"):
        W8A8Linear(128, 8, device="cpu")
    with pytest.raises(RuntimeError, match="HBM"):
        ops.sq_quant_act(torch.zeros(2, 16), None, 1.0, 0, 16)
