"""Auto classes that quantise while loading (reference neural_compressor/transformers/models/modeling_auto.py).

  _BaseINCAutoModelClass.from_pretrained :96-246   load_low_bit :249-799   AutoModelForCausalLM :802

    model = AutoModelForCausalLM.from_pretrained(dir, quantization_config=GPTQConfig(...))   # quantise in HBM
    model.save_pretrained(out)                                                                 # == save_low_bit
    model = AutoModelForCausalLM.from_pretrained(out)                                          # reload packed

The float checkpoint is read straight to the MI355X (`device_map` defaults to "cuda"; 288 GB of HBM hold any model the
reference's layer-wise mode was written to stream) and everything after that is the hot path.  A directory whose config
carries a `quantization_config` -- ours, the reference's, AutoGPTQ's or AutoAWQ's -- is opened with packed modules and
never materialises a dense weight.  Only local directories: there is no network.
"""

import copy
import os
import types

import torch

from ...common.utils import logger
from ..quantization.utils import convert_to_quantized_model, save_low_bit
from ..utils import AwqConfig, GPTQConfig, RtnConfig, TeqConfig

_BY_METHOD = {"rtn": RtnConfig, "awq": AwqConfig, "teq": TeqConfig, "gptq": GPTQConfig}


def _device_of(device_map):
    if isinstance(device_map, dict):  # lm-eval passes {"": device} (reference :299)
        device_map = device_map.get("", "cuda")
    if device_map in (None, "auto"):
        device_map = "cuda"
    dev = torch.device(device_map) if not isinstance(device_map, int) else torch.device("cuda", device_map)
    if dev.type != "cuda":
        raise RuntimeError(f"neural_compressor_amd runs on MI355X only: device_map={device_map!r} is not a HIP device")
    return dev


class _BaseINCAutoModelClass:
    ORIG_MODEL = None

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, **kwargs):
        from transformers import AutoConfig, PretrainedConfig

        device = _device_of(kwargs.pop("device_map", "cuda"))
        config = kwargs.pop("config", None)
        quantization_config = kwargs.pop("quantization_config", None)
        for_inference = kwargs.pop("for_inference", True)
        if not isinstance(config, PretrainedConfig):
            config = AutoConfig.from_pretrained(
                pretrained_model_name_or_path, **{k: v for k, v in kwargs.items() if k in ("trust_remote_code", "revision")}
            )
        saved = getattr(config, "quantization_config", None)
        if saved is None and os.path.isfile(os.path.join(str(pretrained_model_name_or_path), "quantize_config.json")):
            saved = True  # AutoGPTQ-style directory: the settings live next to the weights only
        if saved is not None:
            logger.info("quantization_config: %s", saved)
            model = cls.load_low_bit(pretrained_model_name_or_path, *model_args, config=config, device_map=device, **kwargs)
            logger.info("Saved low bit model loading successfully. Other input args will be ignored.")
            return model

        if isinstance(quantization_config, (RtnConfig, AwqConfig, TeqConfig, GPTQConfig)):
            logger.info("Applying Weight Only Quantization.")
            quantization_config.post_init()
            kwargs.setdefault("torch_dtype", torch.float16 if quantization_config.compute_dtype == "fp16" else torch.bfloat16)
            kwargs["dtype"] = kwargs.pop("torch_dtype")
            model = cls.ORIG_MODEL.from_pretrained(pretrained_model_name_or_path, *model_args, config=config, **kwargs)
            model.eval()
            model.to(device)
            quantization_config.update(device=str(device))
            model = convert_to_quantized_model(model, quantization_config, device=device, for_inference=for_inference)
            if isinstance(quantization_config, AwqConfig):
                quantization_config.backend = "inc"  # reference :218-219: marks the words as already in optimum order
            quantization_config.remove_redundant_parameters()
            model.config.quantization_config = quantization_config
        else:
            if "torch_dtype" in kwargs:
                kwargs["dtype"] = kwargs.pop("torch_dtype")
            model = cls.ORIG_MODEL.from_pretrained(pretrained_model_name_or_path, *model_args, config=config, **kwargs)
            model.eval()
            model.to(device)
        model.device_map = device
        if hasattr(model, "hf_device_map"):
            model.hf_device_map = {"": device}
        model.quantization_config = quantization_config
        if quantization_config is not None:
            model.save_pretrained = types.MethodType(save_low_bit, model)
        logger.info("WeightOnlyQuant done.")
        return model

    @classmethod
    def load_low_bit(cls, pretrained_model_name_or_path, *model_args, **kwargs):
        """Open a saved low-bit checkpoint with packed modules (reference :249-799, minus hub / IPEX handling)."""
        from ...torch.algorithms.weight_only.save_load import load

        device = _device_of(kwargs.pop("device_map", "cuda"))
        config = kwargs.pop("config", None)
        qcfg = copy.deepcopy(getattr(config, "quantization_config", None)) if config is not None else None
        model = load(pretrained_model_name_or_path, format="huggingface", device=str(device), model_class=cls.ORIG_MODEL,
                     **{k: v for k, v in kwargs.items() if k in ("trust_remote_code", "revision")})
        if qcfg is None:
            qcfg = getattr(model.config, "quantization_config", None)
        if qcfg is not None and not isinstance(qcfg, dict):
            qcfg = qcfg.to_dict()
        if isinstance(qcfg, dict):
            method = qcfg.get("quant_method", "gptq")
            method = getattr(method, "value", method)
            quantization_config = _BY_METHOD.get(method, GPTQConfig).from_dict(qcfg)
        else:
            quantization_config = GPTQConfig.from_pretrained(str(pretrained_model_name_or_path))
        quantization_config.remove_redundant_parameters()
        model.config.quantization_config = quantization_config
        model.quantization_config = quantization_config
        model.device_map = device
        model.save_pretrained = types.MethodType(save_low_bit, model)
        return model


def _orig(name):
    import transformers

    return getattr(transformers, name)


class AutoModelForCausalLM(_BaseINCAutoModelClass):
    ORIG_MODEL = _orig("AutoModelForCausalLM")


class AutoModel(_BaseINCAutoModelClass):
    ORIG_MODEL = _orig("AutoModel")


class AutoModelForSeq2SeqLM(_BaseINCAutoModelClass):
    ORIG_MODEL = _orig("AutoModelForSeq2SeqLM")
