from .base_config import BaseConfig, ComposableConfig, ConfigRegistry, config_registry, register_config
from .utils import AWQ, DEFAULT_WHITE_LIST, EMPTY_WHITE_LIST, GPTQ, RTN, Mode, logger

__all__ = [
    "BaseConfig", "ComposableConfig", "ConfigRegistry", "config_registry", "register_config",
    "Mode", "logger", "RTN", "GPTQ", "AWQ", "DEFAULT_WHITE_LIST", "EMPTY_WHITE_LIST",
]
