"""SmoothQuantQuantizer: the algorithm plug-in of SmoothQuant W8A8 (reference smooth_quant/smooth_quant.py:50-245).

prepare()  installs the per-channel min / max observers on every nn.Linear (HIP kernel, statistics stay in HBM);
           the user then runs calibration data through the model;
convert()  removes the observers, computes and applies the smoothing scales (`TorchSmoothQuant.transform`) and replaces
           every selected Linear by `W8A8Linear` (per-channel int8 weights, static per-tensor uint8 activations from the
           same calibration statistics).
The reference's prepare / convert hand the model to `ipex.quantization.prepare / convert`; nothing of that is needed
here, so `example_inputs` is accepted for signature compatibility and otherwise unused.
"""

import torch

from ....common.utils import logger
from ...utils.utility import get_module, set_module
from ..base_algorithm import Quantizer
from .utility import Calibration, SQLinearWrapper, TorchSmoothQuant, W8A8Linear


class SmoothQuantQuantizer(Quantizer):
    def __init__(self, quant_config=None):
        super().__init__(quant_config)

    def _selected(self, model):
        """Names of the nn.Linear layers the per-op config selects (dtype fp32 = leave in floating point)."""
        names = []
        for name, m in model.named_modules():
            if not isinstance(m, torch.nn.Linear):
                continue
            cfg = self.quant_config.get(name) if isinstance(self.quant_config, dict) else None
            if cfg is None or cfg.get("w_dtype") == "fp32":
                continue
            names.append(name)
        return names

    @torch.no_grad()
    def prepare(self, model, example_inputs=None, inplace=True, *args, **kwargs):
        if getattr(model, "_smoothquant_optimized", False):
            logger.info("The model is already optimized by SmoothQuant algorithm, skip it.")
            return model
        dev = torch.device("cuda", torch.cuda.current_device())
        model.to(dev)
        model.eval()
        calib = Calibration(model)
        modules = {n: get_module(model, n) for n in self._selected(model)}
        calib._add_min_max_observer(modules)
        model._sq_calibration = calib
        return model

    @torch.no_grad()
    def convert(self, model, example_inputs=None, inplace=True, *args, **kwargs):
        calib = getattr(model, "_sq_calibration", None)
        assert calib is not None, "convert() needs a model returned by prepare() and run through calibration data"
        calib._remove_observer()
        del model._sq_calibration
        names = [n for n in self._selected(model) if n in calib.input_maxes]
        if not names:
            logger.warning("SmoothQuant: no calibrated Linear layer, the model is returned unchanged")
            return model
        first = self.quant_config[names[0]]
        sq = TorchSmoothQuant(model, scale_sharing=first.get("scale_sharing", False))
        sq.input_mins = {n: calib.input_mins[n] for n in names}
        sq.input_maxes = {n: calib.input_maxes[n] for n in names}
        sq.same_input = {n: o for n, o in calib.same_input.items() if n in sq.input_maxes and o in sq.input_maxes}
        sq.producer = {n: p for n, p in calib.producer.items() if n in sq.input_maxes}
        sq.example_call = calib.example_call  # folds are verified numerically on the first calibration batch
        sq.calibration = calib                # alpha="auto" replays the first calibration forwards (calib.captured_calls)
        sq.transform(alpha=first.get("alpha", 0.5), folding=first.get("folding", False), op_types=(torch.nn.Linear,),
                     scale_sharing=first.get("scale_sharing", False), absorb_to_layer=first.get("absorb_to_layer"),
                     auto_alpha_args=first.get("auto_alpha_args"))
        dev = next(model.parameters()).device
        for name in names:
            mod = get_module(model, name)
            assert isinstance(mod, (torch.nn.Linear, SQLinearWrapper)), type(mod)
            folded = not isinstance(mod, SQLinearWrapper) and name in sq.weight_scale_info
            stat_scale = (1.0 / sq.weight_scale_info[name]) if folded else None  # the producer now emits x / s
            new = W8A8Linear.from_float(mod, calib.input_mins[name], calib.input_maxes[name], device=dev, stat_scale=stat_scale)
            set_module(model, name, new)
        model.sq_info = {"alpha": sq.alpha if isinstance(sq.alpha, dict) else first.get("alpha", 0.5), "folding": first.get("folding", False),
                         "absorb_to_layer": sq.absorb_to_layer}
        logger.info("Smooth quantization done.")
        from types import MethodType

        from .save_load import save

        model.save = MethodType(save, model)  # reference smooth_quant.py:139-141
        return model
