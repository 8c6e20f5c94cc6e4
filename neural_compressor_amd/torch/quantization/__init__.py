from . import algorithm_entry  # registers "rtn" / "gptq" / "awq"  # noqa: F401
from .config import (
    AWQConfig, GPTQConfig, RTNConfig, SmoothQuantConfig, get_default_awq_config, get_default_gptq_config,
    get_default_rtn_config, get_default_sq_config,
)
from .quantize import convert, prepare, quantize
from ..algorithms.weight_only.save_load import load, save  # reference torch/quantization/save_load_entry.py

__all__ = [
    "prepare", "convert", "quantize", "save", "load", "RTNConfig", "GPTQConfig", "AWQConfig", "SmoothQuantConfig",
    "get_default_sq_config",
    "get_default_rtn_config", "get_default_gptq_config", "get_default_awq_config",
]
