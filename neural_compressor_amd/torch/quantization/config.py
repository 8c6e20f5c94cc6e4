"""RTNConfig / GPTQConfig / AWQConfig with the reference's field names and defaults.

Reference: neural_compressor/torch/quantization/config.py -- RTNConfig :119-300, GPTQConfig :322-510,
AWQConfig :525-690.  Only the weight-only-quant configs exist here (SURVEY.md section 8 scope).
Field names, order and default values are kept verbatim so `GPTQConfig(bits=4, group_size=128, ...)` written
for the reference constructs the same object here; lm_head is excluded unless `quant_lm_head` (reference
config.py:242-245 via LM_HEAD_NAMES).
"""

from typing import List, Optional

import torch

from ...common.base_config import BaseConfig, register_config
from ...common.utils import AWQ, DEFAULT_WHITE_LIST, GPTQ, RTN, SMOOTH_QUANT
from ..utils.utility import LM_HEAD_NAMES, PRIORITY_AWQ, PRIORITY_GPTQ, PRIORITY_RTN, PRIORITY_SMOOTH_QUANT, WOQ_WHITE_LIST

FRAMEWORK_NAME = "torch"

__all__ = [
    "RTNConfig", "GPTQConfig", "AWQConfig", "get_default_rtn_config", "get_default_gptq_config",
    "get_default_awq_config", "SmoothQuantConfig", "get_default_sq_config", "FRAMEWORK_NAME",
]


class TorchBaseConfig(BaseConfig):
    """Weight-only configs see every nn.Linear / transformers.Conv1D of the model."""

    def __init__(self, white_list=DEFAULT_WHITE_LIST):
        super().__init__(white_list=white_list)
        object.__setattr__(self, "params_list", self.__class__._generate_params_list())

    @staticmethod
    def get_model_info(model: torch.nn.Module):
        return [(name, type(m).__name__) for name, m in model.named_modules() if isinstance(m, WOQ_WHITE_LIST)]

    @classmethod
    def register_supported_configs(cls):
        cls.supported_configs = []

    def _fp32_for_lm_head(self, **extra):
        """Local override that leaves lm_head in floating point (dtype="fp32" => skipped by every algorithm)."""
        return self.__class__(dtype="fp32", **extra)


def _assign(self, local_vars, names):
    for n in names:
        setattr(self, n, local_vars[n])


@register_config(framework_name=FRAMEWORK_NAME, algo_name=RTN, priority=PRIORITY_RTN)
class RTNConfig(TorchBaseConfig):
    """Round-to-nearest weight-only quantization (reference config.py:119)."""

    name = RTN
    supported_configs: List = []

    def __init__(
        self,
        dtype: str = "int",
        bits: int = 4,
        use_sym: bool = True,
        group_size: int = 32,
        group_dim: int = 1,
        use_full_range: bool = False,
        use_mse_search: bool = False,
        use_layer_wise: bool = True,
        model_path: str = "",
        use_double_quant: bool = False,
        double_quant_dtype: str = "int",
        double_quant_bits: int = 8,
        double_quant_use_sym: bool = False,
        double_quant_group_size: int = 256,
        quant_lm_head: bool = False,
        white_list: Optional[List] = DEFAULT_WHITE_LIST,
        **kwargs,
    ):
        super().__init__(white_list=white_list)
        _assign(self, locals(), [
            "dtype", "bits", "use_sym", "group_size", "group_dim", "use_full_range", "use_mse_search",
            "use_layer_wise", "model_path", "use_double_quant", "double_quant_bits", "double_quant_dtype",
            "double_quant_use_sym", "double_quant_group_size", "quant_lm_head",
        ])
        self._post_init()

    def to_config_mapping(self, config_list=None, model_info=None):
        if not self.quant_lm_head:
            self.set_local(LM_HEAD_NAMES, self._fp32_for_lm_head(use_layer_wise=self.use_layer_wise, model_path=self.model_path))
        return super().to_config_mapping(config_list, model_info)

    @classmethod
    def get_config_set_for_tuning(cls):
        return RTNConfig(dtype=["int4", "nf4"], use_sym=[True, False], group_size=[32, 128], use_mse_search=[False, True])


def get_default_rtn_config(processor_type=None) -> RTNConfig:
    # the reference picks use_layer_wise by CPU brand (config.py:278-288); HBM is 288 GB here: never layer-wise
    return RTNConfig(use_layer_wise=False)


@register_config(framework_name=FRAMEWORK_NAME, algo_name=GPTQ, priority=PRIORITY_GPTQ)
class GPTQConfig(TorchBaseConfig):
    """GPTQ (reference config.py:322)."""

    name = GPTQ
    supported_configs: List = []

    def __init__(
        self,
        dtype: str = "int",
        bits: int = 4,
        use_sym: bool = True,
        group_size: int = 32,
        use_mse_search: bool = False,
        use_layer_wise: bool = False,
        use_block_wise: bool = False,
        model_path: str = "",
        use_double_quant: bool = False,
        double_quant_dtype: str = "int",
        double_quant_bits: int = 8,
        double_quant_use_sym: bool = False,
        double_quant_group_size: int = 256,
        quant_lm_head: bool = False,
        act_order: bool = False,
        hybrid_order: bool = False,
        fp8_aware: bool = False,
        percdamp: float = 0.01,
        block_size: int = 2048,
        static_groups: bool = False,
        true_sequential: bool = False,
        white_list: Optional[List] = DEFAULT_WHITE_LIST,
        **kwargs,
    ):
        super().__init__(white_list=white_list)
        _assign(self, locals(), [
            "dtype", "bits", "use_sym", "group_size", "use_mse_search", "use_layer_wise", "use_block_wise",
            "model_path", "use_double_quant", "double_quant_bits", "double_quant_dtype", "double_quant_use_sym",
            "double_quant_group_size", "act_order", "hybrid_order", "fp8_aware", "percdamp", "block_size",
            "static_groups", "true_sequential", "quant_lm_head",
        ])
        self._post_init()

    def to_config_mapping(self, config_list=None, model_info=None):
        if not self.quant_lm_head:
            self.set_local(
                LM_HEAD_NAMES,
                self._fp32_for_lm_head(use_layer_wise=self.use_layer_wise, model_path=self.model_path, use_block_wise=self.use_block_wise),
            )
        return super().to_config_mapping(config_list, model_info)

    @classmethod
    def get_config_set_for_tuning(cls):
        return GPTQConfig(act_order=[True, False], use_sym=[False, True])


def get_default_gptq_config(processor_type=None) -> GPTQConfig:
    return GPTQConfig()


@register_config(framework_name=FRAMEWORK_NAME, algo_name=AWQ, priority=PRIORITY_AWQ)
class AWQConfig(TorchBaseConfig):
    """AWQ (reference config.py:525)."""

    name = AWQ
    supported_configs: List = []

    def __init__(
        self,
        dtype: str = "int",
        bits: int = 4,
        use_sym: bool = True,
        group_size: int = 32,
        group_dim: int = 1,
        use_full_range: bool = False,
        use_mse_search: bool = False,
        use_layer_wise: bool = False,
        model_path: str = "",
        use_double_quant: bool = False,
        double_quant_dtype: str = "int",
        double_quant_bits: int = 8,
        double_quant_use_sym: bool = True,
        double_quant_group_size: int = 256,
        quant_lm_head: bool = False,
        use_auto_scale: bool = True,
        use_auto_clip: bool = True,
        folding: bool = False,
        white_list: Optional[List] = DEFAULT_WHITE_LIST,
        absorb_layer_dict: dict = {},
        **kwargs,
    ):
        super().__init__(white_list=white_list)
        _assign(self, locals(), [
            "dtype", "bits", "use_sym", "group_size", "group_dim", "use_full_range", "use_mse_search",
            "use_layer_wise", "model_path", "use_double_quant", "double_quant_bits", "double_quant_dtype",
            "double_quant_use_sym", "double_quant_group_size", "quant_lm_head", "use_auto_scale", "use_auto_clip",
            "folding", "absorb_layer_dict",
        ])
        self._post_init()

    def to_config_mapping(self, config_list=None, model_info=None):
        if not self.quant_lm_head:
            self.set_local(LM_HEAD_NAMES, self._fp32_for_lm_head(use_layer_wise=self.use_layer_wise, model_path=self.model_path))
        return super().to_config_mapping(config_list, model_info)

    @classmethod
    def get_config_set_for_tuning(cls):
        return AWQConfig(bits=[4, 6])


def get_default_awq_config() -> AWQConfig:
    return AWQConfig()


@register_config(framework_name=FRAMEWORK_NAME, algo_name=SMOOTH_QUANT, priority=PRIORITY_SMOOTH_QUANT)
class SmoothQuantConfig(TorchBaseConfig):
    """SmoothQuant W8A8 (reference config.py:1485-1612): same fields and defaults.  On MI355X the supported cell is
    the default one -- int8 per-channel symmetric weights, uint8 per-tensor asymmetric min/max activations; `alpha` is a
    number or "auto" (reference smooth_quant/utility.py:1232 AutoAlpha: the layer-wise tuner, or the block-wise one with `do_blockwise=True`)."""

    name = SMOOTH_QUANT
    supported_configs: List = []

    def __init__(
        self,
        w_dtype: str = "int8",
        w_sym: bool = True,
        w_granularity: str = "per_channel",
        w_algo: str = "minmax",
        act_dtype: str = "uint8",
        act_sym: bool = False,
        act_granularity: str = "per_tensor",
        act_algo: str = "minmax",
        excluded_precisions: list = [],
        alpha: float = 0.5,
        folding: bool = False,
        scale_sharing: bool = False,
        init_alpha: float = 0.5,
        alpha_min: float = 0.0,
        alpha_max: float = 1.0,
        alpha_step: float = 0.1,
        shared_criterion: str = "max",
        do_blockwise: bool = False,
        auto_alpha_args: dict = None,
        white_list: Optional[List] = DEFAULT_WHITE_LIST,
        **kwargs,
    ):
        super().__init__(white_list=white_list)
        _assign(self, locals(), [
            "w_dtype", "w_sym", "w_granularity", "w_algo", "act_dtype", "act_sym", "act_granularity", "act_algo",
            "excluded_precisions", "alpha", "folding", "scale_sharing", "init_alpha", "alpha_min", "alpha_max",
            "alpha_step", "shared_criterion", "do_blockwise",
        ])
        self.auto_alpha_args = {
            "init_alpha": init_alpha, "alpha_min": alpha_min, "alpha_max": alpha_max, "alpha_step": alpha_step,
            "shared_criterion": shared_criterion, "do_blockwise": do_blockwise,
        }
        self.absorb_to_layer = kwargs.get("absorb_to_layer", None)
        if w_dtype != "fp32":
            unsupported = []
            if w_dtype != "int8" or not w_sym or w_granularity != "per_channel" or w_algo != "minmax":
                unsupported.append("weights must be int8 / symmetric / per_channel / minmax")
            if act_dtype != "uint8" or act_sym or act_granularity != "per_tensor" or act_algo != "minmax":
                unsupported.append("activations must be uint8 / asymmetric / per_tensor / minmax")
            if unsupported:
                raise NotImplementedError("SmoothQuant on MI355X: " + "; ".join(unsupported))
        self._post_init()

    @staticmethod
    def get_model_info(model: torch.nn.Module, example_inputs=None):
        return [(name, type(m).__name__) for name, m in model.named_modules() if isinstance(m, torch.nn.Linear)]

    @classmethod
    def register_supported_configs(cls):
        cls.supported_configs = []


def get_default_sq_config() -> SmoothQuantConfig:
    return SmoothQuantConfig()
