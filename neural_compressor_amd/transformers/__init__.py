"""transformers-like front-end of the weight-only path.

Reference: neural_compressor/transformers/__init__.py:15-26 -- the same names, so user code written as
    from neural_compressor.transformers import AutoModelForCausalLM, GPTQConfig
    model = AutoModelForCausalLM.from_pretrained(path, quantization_config=GPTQConfig(...))
    model.save_pretrained(out); model = AutoModelForCausalLM.from_pretrained(out)
only changes the package name.  Everything below the front-end is the MI355X hot path (prepare / run / convert).
"""

from .utils import AwqConfig, GPTQConfig, RtnConfig, TeqConfig
from .models import AutoModel, AutoModelForCausalLM, AutoModelForSeq2SeqLM

__all__ = ["RtnConfig", "AwqConfig", "TeqConfig", "GPTQConfig", "AutoModelForCausalLM", "AutoModel", "AutoModelForSeq2SeqLM"]
