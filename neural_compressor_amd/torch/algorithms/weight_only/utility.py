"""Group-wise quantisation helpers with the reference's call signatures, backed by HIP kernels.

Reference: neural_compressor/torch/algorithms/weight_only/utility.py
  quant_tensor         :272-436   -> ops.groupwise_quant (inc_groupwise_quant)
  search_clip          :439-480   -> same grid, loss reduced on device (inc_mse_accumulate)
  model helpers (get_block_prefix :1013, fetch/forward capture :1036-1160) are host-side plumbing.
"""

import torch

from .... import ops
from ....common.utils import logger

__all__ = ["quant_tensor", "search_clip", "get_block_prefix", "get_module", "get_parent"]


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
    if kwargs.get("double_quant", False):
        raise NotImplementedError("double_quant is outside the hot-path scope (SURVEY.md section 8)")
    if dtype != "int":
        raise NotImplementedError(f"dtype={dtype}: only integer formats are in scope (SURVEY.md section 8)")
    if bits <= 0:
        return weight
    assert weight.dim() == 2, "quant_tensor expects a 2-D weight"
    if not weight.is_contiguous():
        raise RuntimeError("quant_tensor works in place and needs a contiguous weight")
    return ops.groupwise_quant(
        weight, bits, group_size, scheme, quantile=quantile, full_range=full_range, return_int=return_int, inplace=True
    )


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
        ops.groupwise_quant(tmp, bits, group_size, scheme, quantile=ratio, full_range=enable_full_range, inplace=True)
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
