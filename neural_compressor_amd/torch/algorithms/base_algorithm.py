"""Algorithm plug-in ABC (interface of the reference's neural_compressor/torch/algorithms/base_algorithm.py:25-126)."""

from abc import ABC, abstractmethod
from collections import OrderedDict

from ...common.utils import Mode


class Quantizer(ABC):
    """prepare -> (user calibration) -> convert; `quantize` chains them; `execute` dispatches on Mode."""

    def __init__(self, quant_config=None, **kwargs):
        self.quant_config = quant_config if quant_config is not None else OrderedDict()

    @abstractmethod
    def prepare(self, model, *args, **kwargs):
        raise NotImplementedError(f"{self.__class__.__name__} must implement `prepare`.")

    @abstractmethod
    def convert(self, model, *args, **kwargs):
        raise NotImplementedError(f"{self.__class__.__name__} must implement `convert`.")

    def quantize(self, model, *args, **kwargs):
        model = self.prepare(model, *args, **kwargs)
        run_fn = kwargs.get("run_fn", None)
        if run_fn is not None:  # RTN has no calibration step
            run_args = kwargs.get("run_args", None)
            if run_args:
                run_fn(model, *run_args)
            else:
                run_fn(model)
        return self.convert(model, *args, **kwargs)

    def execute(self, model, mode, *args, **kwargs):
        if mode == Mode.PREPARE:
            return self.prepare(model, *args, **kwargs)
        if mode == Mode.CONVERT:
            return self.convert(model, *args, **kwargs)
        if mode == Mode.QUANTIZE:
            return self.quantize(model, *args, **kwargs)
        raise NotImplementedError(f"unsupported mode {mode}")
