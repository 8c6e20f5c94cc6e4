from .quantization_config import AwqConfig, GPTQConfig, QuantizationMethod, RtnConfig, TeqConfig, QUANT_CONFIG

__all__ = ["RtnConfig", "AwqConfig", "TeqConfig", "GPTQConfig", "QuantizationMethod", "QUANT_CONFIG"]
