"""neural_compressor_amd: the MI355X-native weight-only-quantization hot path of intel/neural-compressor.

Public surface (same names as the reference's neural_compressor.torch.quantization):
    from neural_compressor_amd.torch.quantization import prepare, convert, quantize, RTNConfig, GPTQConfig, AWQConfig
Importing the package loads libinc_mi355x.so (see _lib.py); a missing extension is a hard error.
"""

__version__ = "0.1.0"

from . import _lib  # noqa: F401  (fails loudly when the HIP extension is absent)
