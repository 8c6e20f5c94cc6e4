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


class HIPAccelerator:
    """The one accelerator of this framework: an MI355X seen through PyTorch-ROCm's `cuda` device type.

    Interface of the reference's Auto_Accelerator (auto_accelerator.py:115-168): name / device_name /
    current_device_name / synchronize / empty_cache / set_device.
    """

    def name(self):
        return "cuda"

    def is_available(self):
        return torch.cuda.is_available()

    def device_name(self, index=None):
        return "cuda" if index is None else f"cuda:{index}"

    def current_device(self):
        return torch.cuda.current_device()

    def current_device_name(self):
        return f"cuda:{torch.cuda.current_device()}"

    def set_device(self, index):
        torch.cuda.set_device(index)

    def synchronize(self):
        torch.cuda.synchronize()

    def empty_cache(self):
        torch.cuda.empty_cache()


_ACC = HIPAccelerator()


def get_accelerator(device_name="auto"):
    """Reference environ.get_accelerator (environ.py:172).  INC_TARGET_DEVICE=cpu is refused: no CPU path here."""
    want = os.environ.get("INC_TARGET_DEVICE", device_name)
    if want not in ("auto", "cuda", None) and not str(want).startswith("cuda"):
        raise RuntimeError(
            f"neural_compressor_amd only drives MI355X GPUs (device type 'cuda' = HIP); requested '{want}'."
        )
    if not _ACC.is_available():
        raise RuntimeError(
            "No HIP device is visible (torch.cuda.is_available() is False). neural_compressor_amd has no CPU fallback."
        )
    return _ACC


def batch_broadcastable(obj):
    """True when `obj` (a block-forward argument other than the hidden states: tensor / nested tuples, lists, dicts /
    plain Python value) can be reused unchanged for a STACK of calibration batches: every tensor in it has a leading
    dimension of 1 (or is 0-d), i.e. it broadcasts over the batch.  Batch-folded arguments such as Bloom / Falcon / MPT
    `alibi` [batch * heads, 1, T] fail this test, and such models are then run one calibration batch per forward, as
    the reference does."""
    import torch

    if isinstance(obj, torch.Tensor):
        return obj.dim() == 0 or obj.shape[0] == 1
    if isinstance(obj, (tuple, list)):
        return all(batch_broadcastable(o) for o in obj)
    if isinstance(obj, dict):
        return all(batch_broadcastable(o) for o in obj.values())
    return True
