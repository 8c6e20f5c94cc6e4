"""Group-wise quantisation helpers with the reference's call signatures, backed by HIP kernels.

Reference: neural_compressor/torch/algorithms/weight_only/utility.py
  quant_tensor         :272-436   -> ops.groupwise_quant (inc_groupwise_quant)
  search_clip          :439-480   -> same grid, loss reduced on device (inc_mse_accumulate)
  model helpers (get_block_prefix :1013, fetch/forward capture :1036-1160) are host-side plumbing.
"""

import torch

from .... import ops
from ....common.utils import logger

__all__ = ["quant_tensor", "quantize_4bit", "search_clip", "get_block_prefix", "get_module", "get_parent",
           "FLOAT_MAPPING", "INT_MAPPING"]

# 4-bit float code books of the reference (utility.py:52-98): QLoRA's NF4 quantiles, the bitsandbytes FP4 table, plain e2m1.
# INT_MAPPING gives the integer stored for each entry (same order; "1111 = -1, 1110 = -2, ..." in the packed nibble).
NF4 = [-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635, -0.18477343022823334,
       -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
       0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0]
FP4_BNB = [-12.0, -8.0, -6.0, -4.0, -3.0, -2.0, -0.0625, 0, 0.0625, 2.0, 3.0, 4.0, 6.0, 8.0, 12.0]
FP4_E2M1 = [-1.0, -0.6666666666666666, -0.5, -0.3333333333333333, -0.25, -0.16666666666666666, -0.010416666666666666, 0.0,
            0.010416666666666666, 0.16666666666666666, 0.25, 0.3333333333333333, 0.5, 0.6666666666666666, 1.0]
NF4_BIT = [7, 1, 2, 3, 4, 5, 6, 0, -8, -7, -6, -5, -4, -3, -2, -1]
FP4_BNB_BIT = [-5, -6, -3, -4, -1, -2, -7, 0, 1, 6, 7, 4, 5, 2, 3]
FP4_E2M1_BIT = [-1, -2, -3, -4, -5, -6, -7, 0, 1, 2, 3, 4, 5, 6, 7]
FLOAT_MAPPING = {"nf4": NF4, "fp4": FP4_BNB, "fp4_e2m1_bnb": FP4_BNB, "fp4_e2m1": FP4_E2M1}
INT_MAPPING = {"nf4": NF4_BIT, "fp4": FP4_BNB_BIT, "fp4_e2m1_bnb": FP4_BNB_BIT, "fp4_e2m1": FP4_E2M1_BIT}


def quantize_4bit(tensor, quantile=1.0, dtype="nf4", return_int=False, **kwargs):
    """NF4 / FP4 quantisation of a [rows, group] tensor, one scale per row (reference utility.py:112-149) -> inc_codebook_quant.
    In place like the reference; return_int (or double_quant) returns (codes-as-ints | quantised tensor, scale [rows,1], None)."""
    assert dtype in FLOAT_MAPPING, "unexpected data type."
    given = kwargs.get("scale") if "scale" in kwargs else None  # utility.py:127-128: the caller's scale replaces the rows' own max
    if return_int or kwargs.get("double_quant", False):
        iw, scale, _ = ops.codebook_quant(tensor, FLOAT_MAPPING[dtype], INT_MAPPING[dtype], -1, quantile=quantile, return_int=True,
                                          scale=given)
        return iw, (given if given is not None else scale), None
    return ops.codebook_quant(tensor, FLOAT_MAPPING[dtype], INT_MAPPING[dtype], -1, quantile=quantile, return_int=False, inplace=True,
                              scale=given)


def quant_tensor(
    weight,
    bits=4,
    group_size=-1,
    scheme="asym",
    quantile=1.0,
    dtype="int",
    return_int=False,
    full_range=False,
    **kwargs,
):
    """Quant(-dequant) a 2-D weight per row in groups of `group_size` (reference utility.py:272).

    In-place like the reference when `return_int=False` (the same storage is returned).  With
    `return_int=True` returns `(int_weight, scale, zp)`; int_weight is an int32 tensor (the reference
    returns the ints in the float storage and casts later, rtn.py:242 / modules.py:331).
    kwargs that are not parameters of this function are swallowed exactly like the reference does
    (this is what makes AWQ's `data_type=` / `num_bits=` calls always quantise as int4).
    """
    if bits <= 0:
        return weight
    assert weight.dim() == 2, "quant_tensor expects a 2-D weight"
    if not weight.is_contiguous():
        raise RuntimeError("quant_tensor works in place and needs a contiguous weight")
    double_quant = bool(kwargs.get("double_quant", False))
    if dtype in FLOAT_MAPPING:  # NF4 / FP4 code books (qdq_weight_actor :262-263)
        def actor(w, want_int):
            return ops.codebook_quant(w, FLOAT_MAPPING[dtype], INT_MAPPING[dtype], group_size, quantile=quantile,
                                      return_int=want_int, inplace=True)
    elif dtype == "int":
        def actor(w, want_int):
            return ops.groupwise_quant(w, bits, group_size, scheme, quantile=quantile, full_range=full_range, return_int=want_int,
                                       inplace=True)
    else:
        raise NotImplementedError(f"dtype={dtype}: integer, NF4 and FP4 weight formats are implemented (fp8 is a plain cast in the reference)")
    if not double_quant:
        return actor(weight, return_int)
    # ---- double quantisation (utility.py:378-436): the [N, G] scales are themselves quantised, as ONE row in groups of
    # double_quant_group_size; "asym" = subtract the mean, quantise symmetrically, add it back
    int_weight, scale, zp = actor(weight, True)
    K_ = weight.shape[1]
    if group_size != -1 and K_ > group_size and K_ % group_size != 0:
        # the reference's "case 3" (a tail group) returns (ints, scale, zp) BEFORE its double-quant step, whatever return_int
        # says (utility.py:334-376): reproduced as it is
        return int_weight, scale, zp
    # the reference double-quantises the scales in the dtype its actor returns them in -- fp32 from qdq_weight_asym, the WEIGHT
    # dtype from qdq_weight_sym and quantize_4bit (probed on the unmodified reference, tests/golden/make_golden_nf4.py): mean,
    # subtraction, quantisation and the final addition all round to it
    if not (dtype == "int" and scheme == "asym"):
        scale = scale.to(weight.dtype)
    scale_dtype = kwargs.get("double_quant_dtype", "int")
    scale_bits = kwargs.get("double_quant_bits", 8)
    scale_scheme = kwargs.get("double_quant_scheme", "asym")
    scale_group_size = kwargs.get("double_quant_group_size", 256)
    if kwargs.get("double_quant_return_int", False):
        # a TODO in the reference (utility.py:383-384): it drops the inner quant_tensor's result and then unpacks the [1, n] scale
        # tensor into three names (:393-405) -- the option has never worked; the same exception, not a different behaviour
        raise ValueError("not enough values to unpack (expected 3, got 1)")
    orig_scale_shape = scale.shape
    flat = scale.reshape(1, -1).contiguous()
    scale_mean = None
    if scale_scheme == "asym":
        scale_mean = flat.mean()
        flat.sub_(scale_mean)
    quant_tensor(flat, dtype=scale_dtype, bits=scale_bits, group_size=scale_group_size, scheme="sym", quantile=1.0,
                 return_int=False, full_range=False)
    if scale_mean is not None:
        flat.add_(scale_mean)
    scale = flat.reshape(orig_scale_shape)
    if return_int:
        return int_weight, scale, zp
    # dequantise with the double-quantised scales, in place (:417-434)
    N, K = weight.shape
    gs = K if (group_size == -1 or K < group_size) else int(group_size)
    vals = int_weight.to(torch.float32)
    if dtype in FLOAT_MAPPING:  # the stored integers index the code book
        lut = torch.zeros(16, dtype=torch.float32, device=weight.device)
        lut[torch.tensor(INT_MAPPING[dtype], device=weight.device) + 8] = torch.tensor(FLOAT_MAPPING[dtype], dtype=torch.float32, device=weight.device)
        vals = lut[(int_weight + 8).long()]
    elif zp is not None:
        vals = vals - zp.repeat_interleave(gs, dim=1)[:, :K]
    weight.copy_((vals * scale.float().repeat_interleave(gs, dim=1)[:, :K]).to(weight.dtype))  # exact product, rounded once = the reference's 16-bit multiply
    return weight


def search_clip(m, bits=4, group_size=32, scheme="asym", dtype="int", enable_full_range=False):
    """Best clip ratio of one Linear by weight MSE over the reference's 40-point grid (utility.py:439-480)."""
    w = m.weight.data
    if not w.is_contiguous():
        w = w.contiguous()
    n_grid, max_shrink = 200, 0.2
    n_try = int(max_shrink * n_grid)
    losses = torch.zeros(n_try, dtype=torch.float64, device=w.device)
    tmp = torch.empty_like(w)
    ratios = []
    for i_s in range(n_try):
        ratio = 1 - i_s / n_grid
        ratios.append(ratio)
        tmp.copy_(w)
        quant_tensor(tmp, bits=bits, group_size=group_size, scheme=scheme, quantile=ratio, dtype=dtype, full_range=enable_full_range)
        ops.mse_accumulate(w, tmp, out=losses[i_s : i_s + 1])
    # first strict minimum, like the reference's `loss < best_error` scan over fp32 means (`.float().pow(2).mean()`):
    # the fp64 fixed-order sums are rounded to fp32 means before they are compared
    vals = (losses / w.numel()).float().tolist()
    best, best_ratio = float("inf"), None
    for ratio, v in zip(ratios, vals):
        if v < best:
            best, best_ratio = v, ratio
    logger.debug("The best clip ratio is %s", best_ratio)
    return best_ratio


# ---------------------------------------------------------------------------------------------------
# model structure helpers (host-side)
# ---------------------------------------------------------------------------------------------------
def get_block_prefix(model):
    """(name of the first nn.ModuleList holding the transformer blocks, number of blocks)."""
    for name, module in model.named_modules():
        if isinstance(module, torch.nn.ModuleList) and len(module) > 0:
            return name, len(module)
    raise ValueError("no torch.nn.ModuleList of blocks found in the model")


def get_module(model, name):
    mod = model
    for part in name.split("."):
        mod = getattr(mod, part)
    return mod


def get_parent(model, name):
    parts = name.split(".")
    return (model if len(parts) == 1 else get_module(model, ".".join(parts[:-1]))), parts[-1]
