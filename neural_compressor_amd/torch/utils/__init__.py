from .utility import (
    LM_HEAD_NAMES, WOQ_WHITE_LIST, algos_mapping, fetch_module, get_accelerator, get_model_device, get_quantizer,
    postprocess_model, register_algo, set_module,
)

__all__ = [
    "LM_HEAD_NAMES", "WOQ_WHITE_LIST", "algos_mapping", "fetch_module", "get_accelerator", "get_model_device",
    "get_quantizer", "postprocess_model", "register_algo", "set_module",
]
