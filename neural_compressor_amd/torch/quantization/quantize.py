"""prepare / convert / quantize: the reference's public entry points (neural_compressor/torch/quantization/quantize.py
:138 quantize, :179 prepare, :253 convert), restricted to the weight-only algorithms."""

import copy

import torch

from ...common.base_config import BaseConfig, ComposableConfig, config_registry
from ...common.utils import Mode, logger
from ..utils.utility import algos_mapping
from .config import FRAMEWORK_NAME


def need_apply(configs_mapping, algo_name):
    return any(cfg.name == algo_name for cfg in configs_mapping.values())


def _as_config(quant_config):
    if isinstance(quant_config, dict):
        registered = config_registry.get_cls_configs()[FRAMEWORK_NAME]
        return ComposableConfig.from_dict(quant_config, config_registry=registered)
    assert isinstance(quant_config, BaseConfig), (
        f"Please pass a dict or config instance as the quantization configuration, but got {type(quant_config)}."
    )
    return quant_config


def preprocess_quant_config(model, quant_config, mode="prepare", example_inputs=None, run_fn=None):
    quant_config = _as_config(quant_config)
    model_info = quant_config.get_model_info(model=model)
    if (getattr(quant_config, "model_path", None) == "" or isinstance(quant_config, ComposableConfig)) and hasattr(model, "name_or_path"):
        quant_config.model_path = model.name_or_path
    return model, quant_config.to_config_mapping(model_info=model_info)


def quantize(model, quant_config, run_fn=None, run_args=None, inplace=True, example_inputs=None):
    """One-shot: prepare -> run_fn(model, *run_args) -> convert."""
    q_model = model if inplace else copy.deepcopy(model)
    q_model, configs_mapping = preprocess_quant_config(q_model, quant_config, mode="quantize", example_inputs=example_inputs, run_fn=run_fn)
    for algo_name, algo_func in algos_mapping.items():
        if need_apply(configs_mapping, algo_name):
            logger.info("Start to apply %s on the model.", algo_name)
            q_model = algo_func(q_model, configs_mapping, run_fn=run_fn, run_args=run_args, example_inputs=example_inputs, mode=Mode.QUANTIZE)
    setattr(q_model, "is_quantized", True)
    return q_model


def prepare(model, quant_config, inplace=True, example_inputs=None, **kwargs):
    """Install the calibration capture of every algorithm the config selects.

    Extra keyword arguments go to the algorithm's `prepare` (an extension of the reference's signature, quantize.py:140): GPTQ takes
    `independent_blocks=True` (one transformer block per rank, calibrated on the float model's activations: distributed.py mode
    "layer"), `hessian_allreduce` / `row_shard_solve` (the exact multi-GPU modes)."""
    prepared = model if inplace else copy.deepcopy(model)
    prepared, configs_mapping = preprocess_quant_config(prepared, quant_config, mode="prepare", example_inputs=example_inputs)
    for algo_name, algo_func in algos_mapping.items():
        if need_apply(configs_mapping, algo_name):
            logger.info("Start to prepare model with %s.", algo_name)
            prepared = algo_func(prepared, configs_mapping, example_inputs=example_inputs, mode=Mode.PREPARE, **kwargs)
            setattr(prepared, "is_prepared", True)
    setattr(prepared, "quant_config", quant_config)
    setattr(prepared, "example_inputs", example_inputs)
    return prepared


def convert(model, quant_config=None, inplace=True, **kwargs):
    """Turn a prepared (and calibrated) model into the packed-weight model."""
    q_model = model if inplace else copy.deepcopy(model)
    is_prepared = getattr(model, "is_prepared", False)
    assert is_prepared or quant_config is not None, "Please pass quant_config to convert function."
    if is_prepared:
        if quant_config is None:
            quant_config = model.quant_config
        else:
            logger.warning("quant_config will be ignored since the model has been prepared.")
            quant_config = model.quant_config
    example_inputs = model.example_inputs if is_prepared else None
    quant_config = _as_config(quant_config)
    configs_mapping = quant_config.to_config_mapping(model_info=quant_config.get_model_info(model=q_model))
    for algo_name, algo_func in algos_mapping.items():
        if need_apply(configs_mapping, algo_name):
            logger.info("Start to convert model with %s.", algo_name)
            q_model = algo_func(q_model, configs_mapping, example_inputs=example_inputs, mode=Mode.CONVERT, **kwargs)
    if hasattr(q_model, "__dict__"):
        setattr(q_model, "is_quantized", True)
    return q_model
