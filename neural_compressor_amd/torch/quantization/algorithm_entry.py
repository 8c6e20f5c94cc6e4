"""Algorithm entries: flatten the per-op config objects into the plain dicts the quantizers take.

Reference: neural_compressor/torch/quantization/algorithm_entry.py -- rtn_entry :63, gptq_entry :121,
awq_quantize_entry :406.  Same registration names ("rtn", "gptq", "awq"), same weight_config keys, same model
attributes afterwards (`model.qconfig`, `model.save`, quantizer parked on `model.quantizer` between phases).
"""

from types import MethodType

import torch

from ...common.utils import AWQ, GPTQ, RTN, SMOOTH_QUANT, Mode, logger
from ..utils.utility import get_quantizer, postprocess_model, register_algo


def _save(self, output_dir="./saved_results", format="default", **kwargs):
    from ..algorithms.weight_only.save_load import save

    return save(self, output_dir, format=format, **kwargs)


@register_algo(RTN)
@torch.no_grad()
def rtn_entry(model, configs_mapping, mode=Mode.QUANTIZE, *args, **kwargs):
    from ..algorithms.weight_only.rtn import RTNQuantizer

    weight_config = {}
    quant_config = None
    for (op_name, op_type), quant_config in configs_mapping.items():
        if quant_config.name != RTN:
            continue
        weight_config[op_name] = {
            "dtype": quant_config.dtype,
            "bits": quant_config.bits,
            "scheme": "sym" if quant_config.use_sym else "asym",
            "group_size": quant_config.group_size,
            "group_dim": quant_config.group_dim,
            "use_full_range": quant_config.use_full_range,
            "use_mse_search": quant_config.use_mse_search,
            "use_double_quant": quant_config.use_double_quant,
            "double_quant_dtype": quant_config.double_quant_dtype,
            "double_quant_bits": quant_config.double_quant_bits,
            "double_quant_scheme": "sym" if quant_config.double_quant_use_sym else "asym",
            "double_quant_group_size": quant_config.double_quant_group_size,
        }
    if quant_config is not None:
        kwargs.update(
            {"use_layer_wise": quant_config.use_layer_wise, "model_path": quant_config.model_path, "quant_lm_head": quant_config.quant_lm_head}
        )
    kwargs.pop("example_inputs", None)
    quantizer = get_quantizer(model, quantizer_cls=RTNQuantizer, quant_config=weight_config)
    model = quantizer.execute(model, mode=mode, *args, **kwargs)
    model.qconfig = configs_mapping
    model.save = MethodType(_save, model)
    postprocess_model(model, mode, quantizer)
    return model


@register_algo(GPTQ)
@torch.no_grad()
def gptq_entry(model, configs_mapping, mode=Mode.QUANTIZE, *args, **kwargs):
    from ..algorithms.weight_only.gptq import GPTQuantizer

    logger.info("Quantize model with the GPTQ algorithm.")
    weight_config = {}
    quant_config = None
    for (op_name, op_type), quant_config in configs_mapping.items():
        if quant_config.name != GPTQ or quant_config.dtype == "fp32":
            continue
        weight_config[op_name] = {
            "dtype": quant_config.dtype,
            "bits": quant_config.bits,
            "sym": quant_config.use_sym,
            "group_size": quant_config.group_size,
            "mse": quant_config.use_mse_search,
            "use_double_quant": quant_config.use_double_quant,
            "double_quant_dtype": quant_config.double_quant_dtype,
            "double_quant_bits": quant_config.double_quant_bits,
            "double_quant_sym": quant_config.double_quant_use_sym,
            "double_quant_group_size": quant_config.double_quant_group_size,
            "act_order": quant_config.act_order,
            "hybrid_order": quant_config.hybrid_order,
            "fp8_aware": quant_config.fp8_aware,
            "percdamp": quant_config.percdamp,
            "block_size": quant_config.block_size,
            "static_groups": quant_config.static_groups,
            "true_sequential": quant_config.true_sequential,
        }
    if quant_config is not None:
        kwargs.update(
            {
                "use_layer_wise": quant_config.use_layer_wise,
                "use_block_wise": quant_config.use_block_wise,
                "model_path": quant_config.model_path,
                "quant_lm_head": quant_config.quant_lm_head,
            }
        )
    kwargs.pop("example_inputs", None)
    quantizer = get_quantizer(model, quantizer_cls=GPTQuantizer, quant_config=weight_config)
    model = quantizer.execute(model, mode=mode, *args, **kwargs)
    model.qconfig = configs_mapping
    model.save = MethodType(_save, model)
    postprocess_model(model, mode, quantizer)
    return model


@register_algo(AWQ)
@torch.no_grad()
def awq_quantize_entry(model, configs_mapping, mode=Mode.QUANTIZE, *args, **kwargs):
    from ..algorithms.weight_only.awq import AWQQuantizer

    logger.info("Quantize model with the AWQ algorithm.")
    weight_config = {}
    use_auto_scale, use_auto_clip, folding, use_full_range, absorb_layer_dict = True, True, False, False, {}
    quant_config = None
    for (op_name, op_type), quant_config in configs_mapping.items():
        if quant_config.name != AWQ:
            continue
        if quant_config.dtype == "fp32":
            weight_config[op_name] = {"bits": -1, "dtype": "fp32", "group_size": 128, "scheme": "asym"}
            continue
        weight_config[op_name] = {
            "dtype": quant_config.dtype,
            "bits": quant_config.bits,
            "group_size": quant_config.group_size,
            "group_dim": quant_config.group_dim,
            "scheme": "sym" if quant_config.use_sym else "asym",
            "use_full_range": quant_config.use_full_range,
            "use_mse_search": quant_config.use_mse_search,
            "use_layer_wise": quant_config.use_layer_wise,
            "use_double_quant": quant_config.use_double_quant,
            "double_quant_dtype": quant_config.double_quant_dtype,
            "double_quant_bits": quant_config.double_quant_bits,
            "double_quant_scheme": quant_config.double_quant_use_sym,
            "double_quant_group_size": quant_config.double_quant_group_size,
        }
        use_auto_scale = quant_config.use_auto_scale
        use_auto_clip = quant_config.use_auto_clip
        folding = quant_config.folding
        use_full_range = quant_config.use_full_range
        absorb_layer_dict = quant_config.absorb_layer_dict
    run_fn = kwargs.get("run_fn", None)
    run_args = kwargs.get("run_args", None)
    example_inputs = kwargs.get("example_inputs", None)
    assert example_inputs is not None, "Please provide example_inputs for AWQ quantization."
    quantizer = get_quantizer(model, quantizer_cls=AWQQuantizer, quant_config=weight_config, absorb_layer_dict=absorb_layer_dict)
    model = quantizer.execute(
        model, mode=mode, bits=-1, example_inputs=example_inputs, run_fn=run_fn, run_args=run_args,
        use_auto_scale=use_auto_scale, use_mse_search=use_auto_clip, folding=folding, use_full_range=use_full_range,
    )
    model.qconfig = configs_mapping
    model.save = MethodType(_save, model)
    postprocess_model(model, mode, quantizer)
    return model


@register_algo(SMOOTH_QUANT)
@torch.no_grad()
def smooth_quant_entry(model, configs_mapping, mode=Mode.QUANTIZE, *args, **kwargs):
    """Reference algorithm_entry.py:285-345 (there: IPEX prepare/convert); here the plain dict per Linear goes to
    SmoothQuantQuantizer and the result carries W8A8Linear modules."""
    from ..algorithms.smooth_quant import SmoothQuantQuantizer

    quant_config = {}
    for (op_name, op_type), cfg in configs_mapping.items():
        if cfg.name != SMOOTH_QUANT:
            continue
        quant_config[op_name] = {
            "w_dtype": cfg.w_dtype, "alpha": cfg.alpha, "folding": cfg.folding, "scale_sharing": cfg.scale_sharing,
            "absorb_to_layer": getattr(cfg, "absorb_to_layer", None), "auto_alpha_args": getattr(cfg, "auto_alpha_args", None),
        }
    run_fn = kwargs.get("run_fn", None)
    example_inputs = kwargs.get("example_inputs", None)
    quantizer = get_quantizer(model, quantizer_cls=SmoothQuantQuantizer, quant_config=quant_config)
    model = quantizer.execute(model, mode=mode, run_fn=run_fn, example_inputs=example_inputs)
    model.qconfig = configs_mapping
    postprocess_model(model, mode, quantizer)
    return model
