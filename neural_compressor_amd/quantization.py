"""2.x-named entry point: `quantization.fit(model, conf, calib_dataloader=...)` (see config.py for why this is a shim)."""

import torch

from .config import PostTrainingQuantConfig
from .torch.quantization import quantize

__all__ = ["fit"]


def fit(model, conf, calib_dataloader=None, calib_func=None, eval_func=None, eval_dataloader=None, eval_metric=None, **kwargs):
    """Weight-only (and SmoothQuant W8A8) post-training quantisation with the 2.x call shape.

    `calib_dataloader` yields model inputs (a tensor, a dict of keyword tensors, or an `(inputs, labels)` pair);
    `calib_func(model)` may be given instead.  Accuracy-driven tuning (`eval_func` & co.) belongs to the 2.x strategy
    layer, which is outside the hot-path scope: the arguments are accepted and ignored.  Returns the quantised
    `torch.nn.Module` (its `.save(dir)` writes the reference's default format).
    """
    if not isinstance(conf, PostTrainingQuantConfig):
        raise TypeError("conf must be a neural_compressor_amd.config.PostTrainingQuantConfig")
    cfg = conf.to_3x()

    def run_fn(m):
        if calib_func is not None:
            return calib_func(m)
        for batch in calib_dataloader:
            if isinstance(batch, (tuple, list)) and len(batch) == 2 and not isinstance(batch[0], (int, float)):
                batch = batch[0]
            if isinstance(batch, dict):
                m(**batch)
            else:
                m(batch)

    needs_calib = cfg.name in ("gptq", "awq", "smooth_quant")
    if needs_calib and calib_dataloader is None and calib_func is None:
        raise ValueError(f"{cfg.name.upper()} needs calibration data: pass calib_dataloader or calib_func")
    example = None
    if cfg.name in ("awq", "smooth_quant"):
        example = kwargs.get("example_inputs")
        if example is None and calib_dataloader is not None:
            first = next(iter(calib_dataloader))
            example = first[0] if isinstance(first, (tuple, list)) else first
    with torch.no_grad():
        return quantize(model, cfg, run_fn=run_fn if needs_calib else None, example_inputs=example)
