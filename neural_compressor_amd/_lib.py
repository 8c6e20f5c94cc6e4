"""ctypes binding of libinc_mi355x.so (the C-ABI declared in include/inc_mi355x.h).

The library is the product: there is NO Python / torch / CPU fallback for any arithmetic on the hot path.
If the shared object is missing or does not export a symbol, importing this module fails loudly.
"""

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libinc_mi355x.so")
ABI_VERSION = 9

INC_OK = 0
INC_F32, INC_F16, INC_BF16 = 0, 1, 2
INC_SCHEME_ASYM, INC_SCHEME_SYM = 0, 1

# name -> (restype, argtypes); mirrors include/inc_mi355x.h one to one
_P = c_void_p
SIGNATURES = {
    "inc_abi_version": (c_int, []),
    "inc_error_string": (c_char_p, [c_int]),
    "inc_target_arch": (c_char_p, []),
    "inc_pack_rows": (c_int, [_P, _P, c_int64, c_int64, c_int, c_int, _P]),
    "inc_unpack_rows": (c_int, [_P, _P, c_int64, c_int64, c_int, c_int, c_int, _P]),
    "inc_woq_pack": (c_int, [_P, c_int, _P, _P, _P, _P, _P, c_int64, c_int64, c_int64, c_int, c_int, _P]),
    "inc_woq_unpack": (c_int, [_P, _P, _P, _P, c_int64, c_int64, c_int64, c_int, _P, _P, _P]),
    "inc_woq_dequant": (c_int, [_P, _P, _P, _P, _P, c_int, c_int64, c_int64, c_int64, c_int, c_int, _P]),
    "inc_dequant_ints": (c_int, [_P, _P, c_int, _P, _P, _P, c_int, c_int64, c_int64, c_int64, c_int, _P]),
    "inc_woq_gemm_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "inc_woq_gemm": (
        c_int,
        [_P, c_int, _P, _P, _P, _P, _P, _P, c_int64, c_int64, c_int64, c_int64, c_int, c_int, _P, c_int64, _P],
    ),
    "inc_woq_gemm_multi_workspace_bytes": (c_int64, [c_int, c_int64, _P, c_int64]),
    "inc_woq_gemm_multi": (c_int, [c_int, _P, c_int, _P, _P, _P, _P, _P, c_int64, _P, c_int64, c_int, c_int, _P, c_int64, _P]),
    "inc_groupwise_quant": (
        c_int,
        [_P, c_int, _P, _P, _P, _P, c_int64, c_int64, c_int, c_int, c_int, c_float, c_int, _P],
    ),
    "inc_codebook_quant": (c_int, [_P, c_int, _P, _P, _P, c_int64, c_int64, c_int, _P, _P, c_int, c_float, _P]),
    "inc_codebook_quant_with_scale": (c_int, [_P, c_int, _P, _P, _P, c_int64, c_int64, c_int, _P, _P, c_int, c_float, _P, _P]),
    "inc_mse_accumulate_workspace_bytes": (c_int64, []),
    "inc_mse_accumulate": (c_int, [_P, _P, c_int, c_int64, _P, _P, _P]),
    "inc_gptq_hessian_accum": (c_int, [_P, c_int, c_int64, c_int64, c_int64, _P, c_float, c_float, _P]),
    "inc_gptq_hessian_accum_multi": (c_int, [c_int, _P, c_int, c_int64, _P, _P, _P, _P, _P, _P, c_int64, _P]),
    "inc_gptq_hessian_accum_multi_workspace_bytes": (c_int64, []),
    "inc_gptq_hessian_finalize": (c_int, [_P, c_int64, c_float, _P, _P, _P]),
    "inc_gptq_prepare_weight": (c_int, [_P, c_int, _P, _P, c_int64, c_int64, _P]),
    "inc_gptq_find_params": (
        c_int,
        [_P, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, _P, _P, c_int64, c_int64, _P],
    ),
    "inc_sq_channel_minmax": (c_int, [_P, c_int, c_int64, c_int64, c_int64, _P, _P, _P]),
    "inc_sq_weight_col_absmax": (c_int, [_P, c_int, c_int64, c_int64, _P, _P]),
    "inc_sq_cal_scale": (c_int, [_P, _P, c_int64, c_float, c_float, _P, _P]),
    "inc_sq_quant_weight": (c_int, [_P, c_int, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P]),
    "inc_sq_quant_act": (c_int, [_P, c_int, c_int64, c_int64, c_int64, _P, c_float, c_float, _P, _P]),
    "inc_w8a8_gemm_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "inc_w8a8_gemm": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int64, c_int64, c_int64, _P, c_int64, _P]),
    "inc_awq_repack": (c_int, [_P, _P, c_int64, c_int64, c_int64, c_int, _P, _P, _P]),
    "inc_gptq_find_params_mse": (
        c_int,
        [_P, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_float, c_float, _P, _P, c_int64, c_int64, _P],
    ),
    "inc_gptq_quant_block": (
        c_int,
        [_P, _P, _P, _P, _P, _P, c_int, _P, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, _P],
    ),
    "inc_gptq_quant_block_params": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, _P]),
    "inc_gptq_lazy_update": (c_int, [_P, _P, _P, c_int64, c_int64, c_int64, c_int, _P]),
    "inc_gptq_lazy_update_cols": (c_int, [_P, _P, _P, c_int64, c_int64, c_int64, c_int, c_int64, c_int64, _P]),
    "inc_probe_hbm_triad": (c_int, [_P, _P, _P, c_float, c_int64, _P]),
    "inc_probe_hbm_copy": (c_int, [_P, _P, c_int64, c_int, _P]),
    "inc_probe_mfma_bf16": (c_int, [_P, _P, c_int, c_int, _P, _P]),
    "inc_trace_marker": (c_int, [c_int, _P]),
    "inc_gptq_quantize_layer": (c_int, [_P, _P, _P, _P, c_int64, _P, _P, c_int64, _P, _P, c_int, _P, c_int64, c_int64, c_int, c_int, c_int,
                                        c_int, c_int, c_int, _P, _P]),
    "inc_chol_diag_block": (c_int, [_P, c_int64, c_int, _P, c_int64, _P, c_int, _P]),
    "inc_gptq_inverse_factor_workspace_bytes": (c_int64, [c_int64, c_int]),
    "inc_gptq_inverse_factor": (c_int, [_P, c_int64, _P, _P, c_int64, _P, c_int, _P, _P]),
    "inc_awq_act_abs_sum": (c_int, [_P, c_int, c_int64, c_int64, _P, _P]),
    "inc_awq_weight_scale_workspace_bytes": (c_int64, [c_int64, c_int64, c_int]),
    "inc_awq_weight_scale": (c_int, [_P, c_int, c_int64, c_int64, c_int, _P, _P, c_int64, _P]),
}


class IncLibraryError(RuntimeError):
    """Raised when libinc_mi355x.so is missing, stale, or a call returns an INC_ERR_* code."""


def _load():
    if not os.path.exists(LIB_PATH):
        raise IncLibraryError(
            f"{LIB_PATH} not found. The MI355X HIP extension is mandatory (there is no CPU fallback): "
            "build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C neural_compressor_amd/csrc`."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise IncLibraryError(f"{LIB_PATH} does not export `{name}`; rebuild the extension") from e
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.inc_abi_version()
    if got != ABI_VERSION:  # pragma: no cover
        raise IncLibraryError(f"ABI mismatch: library reports {got}, bindings expect {ABI_VERSION}; rebuild")
    return lib


lib = _load()


def check(code, what):
    """Turn an INC_ERR_* return code into an exception (the C side never throws)."""
    if code != INC_OK:
        msg = lib.inc_error_string(code).decode()
        raise IncLibraryError(f"{what} failed: {msg} (code {code})")
