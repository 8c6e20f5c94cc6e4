"""RTN: per-Linear group-wise round-to-nearest + pack, all on the MI355X.

Reference: neural_compressor/torch/algorithms/weight_only/rtn.py:68-270 (RTNQuantizer.convert).
Per nn.Linear / transformers.Conv1D found in `quant_config`: optional clip search (utility.search_clip),
`quant_tensor(return_int=True)` -> `MI355XWeightOnlyLinear(...).pack(int_weight, scale, zp, bias)` ->
`set_module`.  Weights are moved to HBM layer by layer; every arithmetic step is a HIP kernel.
"""

import torch

from ....common.utils import logger
from ...utils.utility import WOQ_WHITE_LIST, get_accelerator, get_model_device, set_module
from ..base_algorithm import Quantizer
from .modules import MI355XWeightOnlyLinear
from .utility import FLOAT_MAPPING, quant_tensor, search_clip

try:
    import transformers

    _Conv1D = transformers.Conv1D
except Exception:  # pragma: no cover
    _Conv1D = ()


class RTNQuantizer(Quantizer):
    """`quant_config`: {op_name: {dtype, bits, scheme, group_size, group_dim, use_full_range, use_mse_search, ...}}."""

    def __init__(self, quant_config=None):
        super().__init__(quant_config)

    @torch.no_grad()
    def prepare(self, model, *args, **kwargs):
        return model  # RTN needs no calibration (reference rtn.py:58-66)

    @torch.no_grad()
    def convert(self, model, dtype="int", bits=4, scheme="sym", group_size=32, group_dim=1, quantile=1.0,
                use_full_range=False, use_mse_search=False, use_layer_wise=False, model_path="",
                quant_lm_head=False, *args, **kwargs):
        weight_config = self.quant_config
        device = torch.device(get_accelerator(kwargs.pop("device", "auto")).current_device_name())
        model_device = get_model_device(model)
        use_optimum_format = kwargs.get("use_optimum_format", True)
        if use_layer_wise:
            # the reference's layer-wise mode saves HOST memory by streaming a checkpoint; with 288 GB of HBM the
            # whole model is resident, so the flag is accepted and ignored.
            logger.debug("use_layer_wise is ignored on MI355X")
        assert isinstance(model, torch.nn.Module), "only support torch module"
        for name, m in list(model.named_modules()):
            if not isinstance(m, WOQ_WHITE_LIST) or name not in weight_config:
                continue
            cfg = weight_config[name]
            dtype = cfg.get("dtype", "int")
            if dtype == "fp32":
                continue
            bits = cfg.get("bits", 4)
            if dtype != "int" and "int" in dtype:
                bits = int(dtype.lstrip("int"))
                dtype = "int"
            if dtype != "int" and dtype not in FLOAT_MAPPING:
                # fp8_* is a plain cast in the reference (cast_fp8, rtn.py:167-171), not a weight-only format
                raise NotImplementedError(f"RTN dtype={dtype}: integer, NF4 and FP4 formats are implemented")
            double_quant_config = {  # reference rtn.py:184-190
                "double_quant": cfg.get("use_double_quant", False),
                "double_quant_dtype": cfg.get("double_quant_dtype", "int"),
                "double_quant_bits": cfg.get("double_quant_bits", 8),
                "double_quant_scheme": cfg.get("double_quant_scheme", "asym"),
                "double_quant_group_size": cfg.get("double_quant_group_size", 256),
            }
            group_size = cfg["group_size"]
            scheme = cfg["scheme"]
            quantile = cfg.get("quantile", 1.0)
            group_dim = cfg["group_dim"]
            use_full_range = cfg["use_full_range"]
            use_mse_search = cfg["use_mse_search"]
            logger.debug("RTN %s: bits=%s group_size=%s scheme=%s quantile=%s", name, bits, group_size, scheme, quantile)

            m.to(device)
            is_conv1d = isinstance(m, _Conv1D) if _Conv1D else False
            transpose = (group_dim == 0) ^ is_conv1d  # reference rtn.py:208-214
            weight = m.weight.detach()
            weight = weight.T.contiguous() if transpose else weight.contiguous()
            if use_mse_search:
                quantile = search_clip(m, bits, group_size, scheme, dtype, use_full_range)
            int_weight, scale, zp = quant_tensor(
                weight, dtype=dtype, bits=bits, group_size=group_size, scheme=scheme, quantile=quantile,
                return_int=True, full_range=use_full_range, **double_quant_config,
            )
            if transpose:
                int_weight = int_weight.t().contiguous()
                scale = scale.t().contiguous()
                zp = zp.t().contiguous() if zp is not None else None
            if is_conv1d:
                in_features, out_features = m.weight.shape[0], m.weight.shape[1]
                int_weight = int_weight.t().contiguous()
                scale = scale.t().contiguous()
                zp = zp.t().contiguous() if zp is not None else None
            else:
                in_features, out_features = m.in_features, m.out_features
            new_module = MI355XWeightOnlyLinear(
                in_features, out_features, dtype=dtype, bits=bits, group_size=group_size, zp=zp is not None,
                bias=m.bias is not None, use_optimum_format=use_optimum_format, device=device,
            )
            new_module.pack(int_weight, scale, zp, m.bias)
            if name == "":
                return new_module
            set_module(model, name, new_module)
        # MI355X-first: the quantised model lives in HBM (the packed modules have no host implementation); the
        # reference moves everything back to `model_device` (rtn.py:262-266).
        if model_device != device:
            logger.info("RTN: moving the quantised model from %s to %s", model_device, device)
        model.to(device)
        return model
