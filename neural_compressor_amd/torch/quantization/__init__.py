from . import algorithm_entry  # registers "rtn" / "gptq" / "awq"  # noqa: F401
from .config import (
    AWQConfig, GPTQConfig, RTNConfig, get_default_awq_config, get_default_gptq_config, get_default_rtn_config,
)
from .quantize import convert, prepare, quantize

__all__ = [
    "prepare", "convert", "quantize", "RTNConfig", "GPTQConfig", "AWQConfig",
    "get_default_rtn_config", "get_default_gptq_config", "get_default_awq_config",
]
