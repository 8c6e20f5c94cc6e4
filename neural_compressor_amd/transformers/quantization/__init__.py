from .utils import convert_to_quantized_model, default_run_fn, save_low_bit

__all__ = ["convert_to_quantized_model", "default_run_fn", "save_low_bit"]
