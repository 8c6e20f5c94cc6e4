"""Registry + module helpers (the subset of the reference's neural_compressor/torch/utils/utility.py the hot path uses).

  register_algo / algos_mapping   reference utility.py:63-82
  fetch_module / set_module       reference utility.py:84-127
  get_quantizer / postprocess_model  reference utility.py:163-201 (quantizer parked on model.quantizer between phases)
  HIP accelerator                 reference auto_accelerator.py:220-268 ("cuda" device strings ARE HIP on ROCm)
"""

import os

import torch

from ...common.utils import Mode, logger

try:  # transformers.Conv1D is quantisable too (reference torch/quantization/config.py:66-71)
    import transformers

    WOQ_WHITE_LIST = (torch.nn.Linear, transformers.Conv1D)
except Exception:  # pragma: no cover
    transformers = None
    WOQ_WHITE_LIST = (torch.nn.Linear,)

LM_HEAD_NAMES = [".*lm_head", ".*output_layer", ".*embed_out"]  # reference torch/utils/constants.py:69
PRIORITY_GPTQ, PRIORITY_RTN, PRIORITY_AWQ = 90, 80, 70  # reference torch/utils/constants.py:45-48
PRIORITY_SMOOTH_QUANT = 60

algos_mapping = {}


def register_algo(name):
    """Register `fn(model, configs_mapping, mode=Mode.X, *args, **kwargs)` under an algorithm name."""

    def deco(fn):
        algos_mapping[name] = fn
        return fn

    return deco


def fetch_module(model, op_name):
    mod = model
    for part in op_name.split("."):
        if not hasattr(mod, part):
            logger.warning("The %s is not present in the model.", op_name)
            return None
        mod = getattr(mod, part)
    return mod


def get_module(model, op_name):
    """Fetch a sub-module by dotted name (reference smooth_quant/utility.py:349-369)."""
    mod = model
    for part in op_name.split("."):
        if part == "":
            continue
        mod = getattr(mod, part)
    return mod


def set_module(model, op_name, new_module):
    parts = op_name.split(".")
    parent = model if len(parts) == 1 else fetch_module(model, ".".join(parts[:-1]))
    if parent is None:
        logger.warning("Setting skipped as the %s is not present in the model.", op_name)
        return None
    setattr(parent, parts[-1], new_module)


get_attr = fetch_module


def set_attr(model, name, value):
    set_module(model, name, value)


def get_quantizer(model, quantizer_cls, quant_config=None, *args, **kwargs):
    if hasattr(model, "quantizer"):
        return model.quantizer
    return quantizer_cls(quant_config=quant_config, *args, **kwargs)


def postprocess_model(model, mode, quantizer):
    if mode == Mode.PREPARE:
        model.quantizer = quantizer
    elif mode in (Mode.CONVERT, Mode.QUANTIZE) and getattr(model, "quantizer", False):
        del model.quantizer


def get_model_device(model):
    for p in model.parameters():
        return p.device
    for b in model.buffers():
        return b.device
    return torch.device("cpu")


from .auto_accelerator import (  # noqa: E402,F401  (the reference exposes these through torch.utils too)
    Auto_Accelerator,
    HIPAccelerator,
    accelerator_registry,
    auto_detect_accelerator,
    register_accelerator,
)


def get_accelerator(device_name="auto"):
    """Reference environ.get_accelerator (environ.py:172): the selected accelerator object.  `INC_TARGET_DEVICE=cpu` /
    `device="cpu"` raise -- there is no CPU path."""
    return auto_detect_accelerator(device_name)


def batch_broadcastable(obj):
    """True when `obj` (a block-forward argument other than the hidden states: tensor / nested tuples, lists, dicts /
    plain Python value) can be reused unchanged for a STACK of calibration batches: every tensor in it has a leading
    dimension of 1 (or is 0-d / 1-d), i.e. it broadcasts over the batch.  Batch-folded arguments such as Bloom / Falcon / MPT
    `alibi` [batch * heads, 1, T] fail this test, and such models are then run one calibration batch per forward, as
    the reference does."""
    import torch

    if isinstance(obj, torch.Tensor):
        # (a 1-D tensor has no batch dimension at all -- e.g. `cache_position` [T] of transformers 4.38+: stacking batches does not
        # change it; the callers' faithful-stacking probe compares a stacked forward with the per-batch ones anyway)
        return obj.dim() <= 1 or obj.shape[0] == 1
    if isinstance(obj, (tuple, list)):
        return all(batch_broadcastable(o) for o in obj)
    if isinstance(obj, dict):
        return all(batch_broadcastable(o) for o in obj.values())
    return True
